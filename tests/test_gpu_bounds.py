"""GPU tier: the bounds-checked debug build of the kernels (brotli_amd/csrc/yeast_bounds.h, tools/build_bounds.sh ->
brotli_amd/libyeast_bounds.so): every data-dependent global-memory access of the decode and check kernels is looked
up in the list of the call's buffers; an access outside all of them is refused and the call fails.  The counterpart
of a sanitizer run for device code.  Each case runs in a child process that loads that library."""
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
LIB = os.path.join(H.ROOT, "brotli_amd", "libyeast_bounds.so")

_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import helpers as H
import random_plans as R
import brotli_amd as ba
from brotli_amd.device import DeviceBatch
assert ba.LIB_PATH.endswith("libyeast_bounds.so"), ba.LIB_PATH
mode = sys.argv[1]
meta4, s4 = H.packed_streams("cfg4_streams")
meta5, s5 = H.packed_streams("cfg5_streams")
meta2, s2 = H.packed_streams("cfg2_streams")
if mode == "clean":
    # every kernel form over vectors that reach every path: crafted good/bad streams, the reference's test data, all
    # transforms and overlapping copies (cfg4), the large-window streams (cfg5), random plans with damaged copies, and
    # a slice of the bench's batch; device-resident with regions exactly as long as the outputs, back to back
    crafted = [bytes.fromhex(v["hex"]) for v in H.load_json("crafted.json").values()]
    tdata = [comp for _, comp, _ in H.testdata_pairs()]
    rnd = [s for s, _ in R.streams(120, 5)]
    import random
    rr = random.Random(9)
    dmg = []
    for s in rnd[:80]:
        b = bytearray(s)
        if rr.random() < 0.3:
            b = b[:rr.randrange(1, len(b) + 1)]
        b[rr.randrange(len(b))] ^= 1 << rr.randrange(8)
        dmg.append(bytes(b))
    cases = crafted + tdata + list(s4) + [s5[1]] + rnd + dmg + list(s2[:64])
    want = [H.oracle_yeast(s) for s in cases]
    forms = [(64, 3, 1), (65, 3, 1), (1, 3, 1), (1, 3, 0), (1, 4, 1), (1, 4, 0), (1, 0, 1), (1, 2, 1), (16, 3, 1)]
    for lanes, merge, half in forms:
        os.environ["YEAST_FORCE_LANES"] = str(lanes)
        os.environ["YEAST_MERGE"] = str(merge)
        os.environ["YEAST_HALF"] = str(half)
        res = ba.yeast_batch(cases, ba.CHECK_XXH64)      # host-pointer path (a refused access fails the call)
        for i, (w, r) in enumerate(zip(want, res)):
            assert (r[0], r[2], r[3], r[1]) == (w[0], w[1], w[2], w[3]), (lanes, merge, half, i)
        good = [(s, w) for s, w in zip(cases, want) if w[0] == 0 and w[1] > 0]
        b = DeviceBatch([s for s, _ in good], [w[1] for _, w in good], align=1)
        b.decode(check=ba.CHECK_CRC32C, lanes=lanes)
        rec = b.records()
        for i, (s, w) in enumerate(good):
            assert (rec[i]["rc"], rec[i]["got"], rec[i]["used"]) == w[:3], (lanes, merge, half, i)
    # the single big stream through yeast() (long copies by a whole wavefront, the retry pass for > 1 MiB)
    for k in range(2):
        rc, out, got, used = ba.yeast(s5[k])
        assert (rc, got) == (meta5["streams"][k]["rc"], meta5["streams"][k]["got"])
    print("bounds clean ok")
else:
    # the checker itself: the output arena declared 8 bytes shorter than it is -> the last stream's last stores are
    # refused, the call fails and says where
    os.environ["YEAST_BOUNDS_SHRINK"] = "8"
    streams = list(s2[:8])
    caps = [m["got"] for m in meta2["streams"][:8]]
    for lanes in (1, 65, 64):
        b = DeviceBatch(streams, caps, align=1)
        try:
            b.decode(check=ba.CHECK_NONE, lanes=lanes)
        except RuntimeError as e:
            msg = str(e)
            assert "outside every buffer" in msg and "d_dst" in msg, msg
        else:
            raise AssertionError("lanes %%d: a store past the declared end of d_dst was not caught" %% lanes)
    os.environ["YEAST_BOUNDS_SHRINK"] = "0"
    b = DeviceBatch(streams, caps, align=1)
    b.decode(check=ba.CHECK_NONE, lanes=1)
    assert (b.records()["rc"] == 0).all()
    print("bounds caught ok")
"""


def _run(mode):
    if not os.path.exists(LIB):
        pytest.fail("%s is missing: run tools/build_bounds.sh (or __graft_entry__.build())" % LIB)
    env = dict(os.environ, YEAST_LIB=LIB)
    out = subprocess.run([sys.executable, "-c", _CHILD % {"root": H.ROOT, "tests": os.path.join(H.ROOT, "tests")}, mode],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and ("bounds %s ok" % mode) in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bounds_checked_build_finds_no_access_outside_the_buffers():
    _run("clean")


def test_bounds_checked_build_catches_a_store_past_the_declared_end():
    _run("caught")

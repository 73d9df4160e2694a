"""The entry point that allocates nothing (include/yeast_batch.h: yeast_batch_device_static) against the
entry point it is a variant of, through the C-ABI: same records, same bytes, on two calls in a row, the second
with a smaller input.  Collected last: a failure here must not hide the other GPU tests."""
import numpy as np
import pytest

import brotli_amd as ba
import helpers as H

pytestmark = pytest.mark.gpu


def test_static_entry_point_matches_the_allocating_one():
    import torch
    from brotli_amd.device import DeviceBatch
    assert ba.lib().yeast_init() == 0
    meta, distinct = H.packed_streams("cfg2_streams")
    _, misc = H.packed_streams("misc_streams")
    crafted = [bytes.fromhex(v["hex"]) for v in H.load_json("crafted.json").values()]
    first = [distinct[(5 * j + 1) % 256] for j in range(700)] + misc + crafted
    second = [distinct[(3 * j + 2) % 256] for j in range(90)] + crafted[:7]
    for streams in (first, second):
        caps = [max(H.oracle_yeast(s)[1], 1) + 64 for s in streams]
        want = DeviceBatch(streams, caps)
        want.dst.fill_(0xAA)
        want.decode(check=ba.CHECK_XXH64, lanes=64)
        have = DeviceBatch(streams, caps)
        have.dst.fill_(0xAA)
        # free device memory as the driver counts it (hipMemGetInfo): an allocation inside the library would not
        # show in torch's own counters
        torch.cuda.synchronize()
        before = torch.cuda.mem_get_info()[0]
        have.decode_static(check=ba.CHECK_XXH64)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == before
        a, b = want.records(), have.records()
        assert a.tobytes() == b.tobytes()
        off = have.host_dst_off
        x, y = want.dst.cpu().numpy(), have.dst.cpu().numpy()
        for i in range(len(streams)):
            g = int(a["got"][i])
            assert np.array_equal(x[int(off[i]):int(off[i]) + g], y[int(off[i]):int(off[i]) + g]), i
    # more streams than the static workspace is laid out for: refused, with a message
    big = DeviceBatch([distinct[0]] * 4, [65536] * 4)
    e = ba.lib().yeast_batch_device_static(big.src.data_ptr(), big.src_off.data_ptr(), big.dst.data_ptr(),
                                           big.dst_off.data_ptr(), big.rec.data_ptr(), 16385, 0, None)
    assert e == -1 and b"at most" in ba.lib().yeast_last_error()


_FIRST_USE = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import brotli_amd as ba, helpers as H
from brotli_amd.device import DeviceBatch
meta, distinct = H.packed_streams("cfg2_streams")
streams = distinct[:24]
torch.cuda.set_device(0)
b = DeviceBatch(streams, [65536] * len(streams))
assert ba.lib().yeast_init() == 0
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
b.decode_static(check=ba.CHECK_XXH64)      # the library's first decode in this process
torch.cuda.synchronize()
assert torch.cuda.mem_get_info()[0] == free0, "the static call allocated device memory"
r = b.records()
for i, m in enumerate(meta["streams"][:24]):
    assert (int(r["rc"][i]), int(r["got"][i]), int(r["used"][i])) == (0, m["got"], m["len"]), i
    assert int(r["check"][i]) == int(m["xxh64"], 16), i
print("first-use ok")
"""


def test_static_entry_point_first_in_a_fresh_process():
    """Straight after yeast_init(), before any allocating call has warmed anything up."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _FIRST_USE % (root, os.path.join(root, "tests"))],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "first-use ok" in out.stdout, out.stdout + out.stderr


def test_static_entry_point_under_graph_capture():
    """Warm up on one stream, capture the static call on another, replay twice: the call must neither wait
    for nor record the library's per-device event while its stream is being captured."""
    import torch
    from brotli_amd.device import DeviceBatch
    assert ba.lib().yeast_init() == 0
    meta, distinct = H.packed_streams("cfg2_streams")
    streams = [distinct[(7 * j + 3) % 256] for j in range(48)]
    want = DeviceBatch(streams, [65536] * len(streams))
    want.decode(check=ba.CHECK_XXH64, lanes=64)          # on the default stream: leaves the event recorded there
    have = DeviceBatch(streams, [65536] * len(streams))
    have.decode_static(check=ba.CHECK_XXH64)             # an ordinary static call first (event on the default stream)
    torch.cuda.synchronize()
    have.rec.zero_()
    have.dst.fill_(0x55)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        have.decode_static(check=ba.CHECK_XXH64)
    torch.cuda.synchronize()
    for _ in range(2):
        have.rec.zero_()
        have.dst.fill_(0x55)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert want.records().tobytes() == have.records().tobytes()
        assert torch.equal(want.dst[:int(want.host_dst_off[-1])], have.dst[:int(have.host_dst_off[-1])])
    # the library's ordinary calls still work after a capture (the event was left alone)
    want.decode(check=ba.CHECK_XXH64, lanes=64)
    with torch.cuda.stream(side):
        have.decode_static(check=ba.CHECK_XXH64)
    torch.cuda.synchronize()
    assert want.records().tobytes() == have.records().tobytes()

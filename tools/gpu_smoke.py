#!/usr/bin/env python3
"""Ad-hoc GPU check used during development: product vs oracle on the golden fixtures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import brotli_amd as ba
import helpers as H

bad = 0
def chk(name, a, b):
    global bad
    if a != b:
        bad += 1
        sa = tuple(x if not isinstance(x, bytes) else ("%d bytes" % len(x)) for x in a)
        sb = tuple(x if not isinstance(x, bytes) else ("%d bytes" % len(x)) for x in b)
        print("MISMATCH", name, "gpu", sa, "oracle", sb, flush=True)
    else:
        print("ok", name, a[:3], flush=True)

print("devices", ba.lib().yeast_device_count(), flush=True)
cr = H.load_json("crafted.json")
for k, v in sorted(cr.items()):
    src = bytes.fromhex(v["hex"])
    rc, out, got, used = ba.yeast(src)
    chk(k, (rc, got, used, out), H.oracle_yeast(src))
for name, comp, orig in H.testdata_pairs():
    t = time.time()
    rc, out, got, used = ba.yeast(comp)
    dt = time.time() - t
    chk(name + " %.1f ms" % (dt * 1e3), (rc, got, used, out), H.oracle_yeast(comp))
    rc, _, got, used = ba.yeast(comp, orig)
    chk(name + " cmp", (rc, got, used, None), H.oracle_yeast(comp, orig))
    if len(orig) > 10:
        e2 = bytearray(orig); e2[len(e2) // 2] ^= 1; e2 = bytes(e2)
        rc, _, got, used = ba.yeast(comp, e2)
        chk(name + " cmp-mismatch", (rc, got, used, None), H.oracle_yeast(comp, e2))
        rc, _, got, used = ba.yeast(comp, orig[:-3])
        chk(name + " cmp-short", (rc, got, used, None), H.oracle_yeast(comp, orig[:-3]))
meta, streams = H.packed_streams("misc_streams")
for m, s in zip(meta["streams"], streams):
    rc, out, got, used = ba.yeast(s)
    chk("misc %s %s" % (m["file"], m["params"]), (rc, got, used, out), H.oracle_yeast(s))
meta, streams = H.packed_streams("cfg2_streams")
for lanes in (65, 64, 16, 8, 1):
    os.environ["YEAST_LANES"] = str(lanes)
    t = time.time()
    res = ba.yeast_batch(streams * 8, ba.CHECK_XXH64)
    dt = time.time() - t
    nb = 0
    for i, r in enumerate(res):
        m = meta["streams"][i % len(streams)]
        if r[0] != 0 or r[2] != m["got"] or r[3] != m["len"] or "%016x" % r[4] != m["xxh64"]:
            nb += 1
    if nb == 0:
        o = H.oracle_yeast(streams[5])
        nb += res[5][1] != o[3]
    print("cfg2 batch lanes=%d: %d streams %.1f ms host-inclusive, %d bad" % (lanes, len(res), dt * 1e3, nb), flush=True)
    bad += nb
kat = H.load_json("checks.json")
alice = open(os.path.join(H.GOLD, "testdata", "alice29.txt"), "rb").read()
for k, v in kat.items():
    data = b"123456789" if k == "123456789" else b"" if k == "empty" else alice if k == "alice29.txt" else alice[:int(k.split(":")[1][:-1])]
    g = dict(crc32c="%08x" % ba.crc32c(data), xxh32="%08x" % ba.xxh32(data), xxh64="%016x" % ba.xxh64(data))
    if g != v:
        bad += 1
        print("CHECK MISMATCH", k, g, v)
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)

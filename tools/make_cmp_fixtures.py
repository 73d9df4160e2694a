#!/usr/bin/env python3
"""Compare-mode fixtures (build container only): what the REAL reference's yeast(cmp=1) returns --
(rc, *got, *len) -- when the expected data differs from the decoded data at a chosen byte, is cut short, or
runs on (yeast.h:18-23; yeast.c:1042-1045, 1216-1219, 1270-1273, 1286-1289).  Only recipes are stored
(which stream, which change): the expected data is rebuilt from the committed originals.
Output: tests/golden/cmp_cases.json.  Needs oracle/_ref/libyeast_ref.so (make -C oracle)."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H


def expected_of(orig: bytes, case):
    op = case["op"]
    if op == "same":
        return orig
    if op == "xor":
        b = bytearray(orig)
        b[case["pos"]] ^= case["mask"]
        return bytes(b)
    if op == "cut":
        return orig[:case["pos"]]
    if op == "more":
        return orig + bytes([case["mask"]]) * case["pos"]
    raise ValueError(op)


def main():
    ref = H.reference()
    assert ref is not None, "build oracle/_ref first"
    rnd = random.Random(0xC0DE)
    sources = []
    for name, comp, orig in H.testdata_pairs():
        sources.append(("testdata/" + name, comp, orig))
    meta, streams = H.packed_streams("cfg2_streams")
    for i in (0, 70, 200, 255):
        sources.append(("cfg2/%d" % i, streams[i], H.run_yeast(ref.yeast, streams[i])[3]))
    meta, streams = H.packed_streams("misc_streams")
    for i in range(0, len(streams), 7):
        sources.append(("misc/%d" % i, streams[i], H.run_yeast(ref.yeast, streams[i])[3]))
    cases = []
    for name, comp, orig in sources:
        n = len(orig)
        todo = [{"op": "same"}]
        if n:
            spots = sorted(set([0, n - 1, n // 2, n // 3] + [rnd.randrange(n) for _ in range(6)]))
            todo += [{"op": "xor", "pos": p, "mask": rnd.choice([1, 0x20, 0x80, 0xff])} for p in spots]
            todo += [{"op": "cut", "pos": p} for p in sorted(set([0, 1, n - 1, n // 2, rnd.randrange(n)])) if p < n]
        todo += [{"op": "more", "pos": k, "mask": 0x41} for k in (1, 1000)]
        for c in todo:
            rc, got, used, _ = H.run_yeast(ref.yeast, comp, expected_of(orig, c))
            c.update({"stream": name, "rc": rc, "got": got, "used": used})
            cases.append(c)
    json.dump({"about": "reference yeast(cmp=1): tools/make_cmp_fixtures.py", "cases": cases},
              open(os.path.join(H.GOLD, "cmp_cases.json"), "w"), indent=0)
    print(len(cases), "cases;", sum(1 for c in cases if c["rc"] == 4), "mismatches")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-off soak: many randomly assembled streams (tests/random_plans.py, several seeds) through the batch kernel
(every loop form, half-filled and full wavefronts), the wavefront-per-stream kernel and the cooperative kernel against
the CPU restatement.
usage: tools/soak_random.py [seeds] [streams per seed] [damaged: 0|1] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import random_plans as R
import brotli_amd as ba
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = 0
first = int(sys.argv[4]) if len(sys.argv) > 4 else 100
for seed in range(first, first + seeds):
    t0 = time.time()
    ss = R.streams(per, seed)
    if len(sys.argv) > 3 and sys.argv[3] != "0":  # damaged copies too: bit flips and cuts anywhere (same rc, got, used and partial output wanted)
        import random
        rnd = random.Random(seed)
        extra = []
        for s, _ in ss:
            b = bytearray(s)
            if rnd.random() < 0.3:
                b = b[:rnd.randrange(1, len(b) + 1)]
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            extra.append((bytes(b), None))
        ss = ss + extra
    want = [H.oracle_yeast(s) for s, _ in ss]
    for lanes, merge, half in ((1, 3, 1), (1, 3, 0), (1, 4, 1), (1, 4, 0), (1, 0, 1), (1, 1, 1), (1, 2, 1), (1, 2, 0), (65, 3, 1), (64, 3, 1)):
        os.environ["YEAST_FORCE_LANES"] = str(lanes)
        os.environ["YEAST_MERGE"] = str(merge)
        os.environ["YEAST_HALF"] = str(half)
        res = ba.yeast_batch([s for s, _ in ss], ba.CHECK_NONE)
        for i, (w, r) in enumerate(zip(want, res)):
            if (r[0], r[2], r[3], r[1]) != (w[0], w[1], w[2], w[3]):
                bad += 1
                print("MISMATCH seed %d stream %d lanes %d merge %d half %d: gpu %s oracle %s" % (seed, i, lanes, merge, half, (r[0], r[2], r[3]), w[:3]), flush=True)
    print("seed %d: %d streams, %.1f s" % (seed, per, time.time() - t0), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Generate tests/golden/cfg4_streams.{bin,json} and cfg5_streams.{bin,json} (BASELINE configs 4, 5).

Streams come from tools/brosynth.py (this repo's stream synthesiser); every expected value (rc,
got, used, check values, the output itself for the small ones) is what the REAL reference decoder
(oracle/_ref/libyeast_ref.so, built from /root/reference by oracle/Makefile) returns for them.  The
synthesiser's own idea of the output is only used as a cross-check.  Runs in the build container.

  cfg4  all 121 transforms x word lengths 4..24, the four context modes with 64-entry context maps,
        NPOSTFIX 0..3 x NDIRECT extremes, block switching (type codes 0 and 1 included), RLEMAX /
        inverse move-to-front context maps, simple code shapes, overlapping copies of every short
        period, uncompressed and metadata meta-blocks.
  cfg5  large-window single streams: window bits 24, back-references 16 MiB away, copies of many
        megabytes.  Only their check values are kept (the outputs are 17-40 MB).
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import ctypes

import brosynth as B
import helpers as H

GOLD = os.path.join(ROOT, "tests", "golden")


def luts():
    L = H.oracle()
    return tuple(ctypes.string_at(L.oracle_context_lut(k), 256) for k in range(3))


def dict_word(copy, ident):
    buf = ctypes.create_string_buffer(64)
    n = H.oracle().oracle_dict_word(buf, copy, ident)
    return None if n < 0 else buf.raw[:n]


def checks(data):
    ref = H.reference()
    return dict(crc32c="%08x" % ref.crc32c(0, data, len(data)), xxh32="%08x" % ref.XXH32(data, len(data), 0),
                xxh64="%016x" % ref.XXH64(data, len(data), 0))


def ref_decode(src):
    return H.run_yeast(H.reference().yeast, src)


TEXT = open(os.path.join(GOLD, "testdata", "alice29.txt"), "rb").read()


def text_cmds(rng, nbytes, maxcopy=40, lit_p=0.5, far=False):
    """a plausible command sequence over English text: literal runs and copies of earlier output"""
    cmds = []
    pos = rng.randrange(0, len(TEXT) - nbytes - 1)
    produced = 0
    total_prev = [0]

    def add(c):
        cmds.append(c)
    while produced < nbytes:
        nl = rng.choice([0, 0, 1, 1, 2, 3, 5, 8, 13, 30]) if rng.random() < lit_p or produced < 8 else 0
        if produced < 8:
            nl = max(nl, 4)
        lits = TEXT[pos:pos + nl]
        pos += nl
        produced += nl
        copy = rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 24, 33, maxcopy])
        kind = rng.random()
        have = produced + total_prev[0]
        if kind < 0.15 and not far:
            dist = ("implicit",)
        elif kind < 0.35:
            dist = ("ring", rng.randrange(16))
        else:
            dist = ("dist", rng.randrange(1, max(2, min(have, 60000))))
        add(dict(lits=lits, copy=copy, dist=dist))
        produced += copy
    cmds.append(dict(lits=TEXT[pos:pos + 3], copy=None))
    return cmds


def fix_ring(cmds, model_start_ring=None):
    """ring codes and implicit distances must resolve to a usable distance: replace those that do
    not by explicit distances (resolved while the plan is replayed on a scratch model)"""
    ring = [16, 15, 11, 4]
    ptr = 3
    out = 0
    res = []
    for c in cmds:
        out += len(c.get("lits", b""))
        if c.get("copy") is None:
            res.append(c)
            break
        d = c["dist"]
        mx = out
        if d[0] == "implicit":
            v = ring[ptr]
            if not (0 < v <= mx):
                d = ("dist", max(1, min(mx, 1 + (v % max(mx, 1)))))
        elif d[0] == "ring":
            v = ring[(ptr - B.RING_BACK[d[1]]) & 3] + B.RING_DELTA[d[1]]
            if not (0 < v <= mx):
                d = ("dist", max(1, min(mx, 1 + (abs(v) % max(mx, 1)))))
        if d[0] == "dist":
            d = ("dist", max(1, min(d[1], mx)))
            val, code = d[1], 16
        elif d[0] == "ring":
            val, code = ring[(ptr - B.RING_BACK[d[1]]) & 3] + B.RING_DELTA[d[1]], d[1]
        else:
            val, code = ring[ptr], 0
        if d[0] != "implicit" and code != 0:
            ptr = (ptr + 1) & 3
            ring[ptr] = val
        c = dict(c)
        c["dist"] = d
        res.append(c)
        out += c["copy"]
    return res


def plans():
    rng = random.Random(0xC0FFEE)
    out = []

    # --- every transform at every word length, three meta-blocks of ~850 words each
    refs = [(L, t) for t in range(121) for L in range(4, 25)]
    rng.shuffle(refs)
    for part in range(3):
        cmds = []
        for k, (L, t) in enumerate(refs[part::3]):
            idx = rng.randrange(1 << B.NDBITS[L])
            lits = bytes([rng.choice(b" ,.;\n")]) if k % 3 == 0 else b""
            cmds.append(dict(lits=lits, copy=L, dist=("dict", L, idx, t)))
        cmds.append(dict(lits=b"end", copy=None))
        # the window must stay empty enough for every id to be reachable: wbits 10 keeps max small
        out.append(("dict_all_%d" % part, dict(wbits=10, blocks=[("c", dict(cmds=cmds, lit_trees=1))]), part))

    # --- the four context modes, 64-entry context maps with many trees, all map styles
    for mode in range(4):
        for trees, style in ((2, dict(rlemax=0, imtf=False)), (7, dict(rlemax=3, imtf=True)), (23, dict(rlemax=16, imtf=False)),
                             (64, dict(rlemax=6, imtf=True))):
            cmap = [rng.randrange(trees) for _ in range(64)]
            for i in range(trees):  # every tree is used
                cmap[i % 64] = i
            cmds = fix_ring(text_cmds(rng, 6000, lit_p=0.9))
            mb = dict(cmds=cmds, modes=[mode], lit_trees=trees, lit_cmap=cmap, style=dict(cmapL=style))
            out.append(("ctxmode%d_trees%d" % (mode, trees), dict(wbits=16, blocks=[("c", mb)]), mode * 100 + trees))

    # --- NPOSTFIX x NDIRECT, with distance context maps
    for P in range(4):
        for nd in (0, 1, 7, 15):
            ND = nd << P
            dist_trees = rng.choice([1, 2, 4])
            dmap = [rng.randrange(dist_trees) for _ in range(4)]
            cmds = fix_ring(text_cmds(rng, 20000, lit_p=0.3))
            mb = dict(cmds=cmds, npostfix=P, ndirect=ND, dist_trees=dist_trees, dist_cmap=dmap,
                      style=dict(cmapD=dict(rlemax=rng.choice([0, 1, 2]), imtf=rng.random() < 0.5)))
            out.append(("postfix%d_direct%d" % (P, ND), dict(wbits=18, blocks=[("c", mb)]), P * 1000 + ND))

    # --- block switching in all three categories, type codes 0 (previous) and 1 (next) included
    for nL, nI, nD in ((2, 1, 1), (1, 3, 1), (1, 1, 4), (5, 4, 3), (17, 9, 6), (256, 2, 2)):
        def sched(n, unit):
            seg = [(0, rng.randrange(1, unit))]
            for _ in range(60):
                seg.append((rng.randrange(n), rng.choice([1, 1, 2, 5, 17, 40, 100, 400])))
            return seg
        nLc = min(nL, 40)
        trees = min(nL * 2, 40)
        cmap = [rng.randrange(trees) for _ in range(64 * nL)]
        dist_trees = min(nD * 2, 8)
        dmap = [rng.randrange(dist_trees) for _ in range(4 * nD)]
        cmds = fix_ring(text_cmds(rng, 30000, lit_p=0.6))
        mb = dict(cmds=cmds, ntypes=dict(L=nL, I=nI, D=nD),
                  sched=dict(L=sched(nL, 50), I=sched(nI, 20), D=sched(nD, 20)),
                  modes=[rng.randrange(4) for _ in range(nL)], lit_trees=trees, lit_cmap=cmap,
                  dist_trees=dist_trees, dist_cmap=dmap)
        out.append(("blocks_%d_%d_%d" % (nL, nI, nD), dict(wbits=17, blocks=[("c", mb)]), nL * 7 + nI * 3 + nD))
        del nLc

    # --- simple code shapes and single-symbol codes: tiny alphabets in use
    for k in range(8):
        alpha = bytes(rng.sample(range(32, 127), rng.choice([1, 2, 3, 4])))
        cmds = []
        for _ in range(40):
            cmds.append(dict(lits=bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 5))), copy=rng.choice([2, 3]),
                             dist=("dist", 1)))
        cmds[0]["lits"] = bytes([alpha[0]]) + cmds[0]["lits"]
        cmds.append(dict(lits=bytes([alpha[0]]), copy=None))
        mb = dict(cmds=fix_ring(cmds), style=dict(lit=dict(simple=True, shape5=k % 2 == 0, pad_simple=k % 3 == 0),
                                                  iac=dict(simple=k % 2 == 1), dist=dict(simple=True)))
        out.append(("simple_%d" % k, dict(wbits=10 + k, blocks=[("c", mb)]), k))

    # --- overlapping copies: every period 1..40, short and long, then non-overlapping long ones
    cmds = [dict(lits=TEXT[1000:1050], copy=5, dist=("dist", 7))]
    for d in range(1, 41):
        for n in (d + 1, 2 * d + 3, 31, 32, 33, 64, 65, 100, 257):
            if n > 2:
                cmds.append(dict(lits=TEXT[2000 + d:2000 + d + (d % 3)], copy=n, dist=("dist", d)))
    for n in (24, 31, 32, 33, 63, 64, 65, 500, 2000):
        cmds.append(dict(lits=b"", copy=n, dist=("dist", 4000)))
    cmds.append(dict(lits=b".", copy=None))
    out.append(("overlaps", dict(wbits=16, blocks=[("c", dict(cmds=fix_ring(cmds)))]), 77))

    # --- several meta-blocks: compressed, uncompressed, metadata, empty last
    blocks = [("c", dict(cmds=fix_ring(text_cmds(rng, 3000)))), ("u", TEXT[5000:5700]), ("m", b"metadata bytes"),
              ("m", b""), ("c", dict(cmds=[dict(lits=b"", copy=30, dist=("dist", 650)), dict(lits=b"tail", copy=None)],
                                     lit_trees=2, lit_cmap=[i & 1 for i in range(64)], modes=[2])),
              ("u", TEXT[9000:9001]), ("e",)]
    out.append(("mixed_blocks", dict(wbits=22, blocks=blocks), 5))
    return out


def far_plans():
    """cfg5: window bits 24, 16 MiB back-references"""
    res = []
    seed_text = TEXT[:65536]
    W = (1 << 24) - 16
    # 64 KiB stored, doubled by copies up to 16 MiB, then references at the far edge of the window
    blocks = [("u", seed_text)]
    size = 65536
    cmds = []
    while size < (1 << 24):
        n = min(size, (1 << 24) - size)
        cmds.append(dict(lits=b"", copy=n, dist=("dist", size)))
        size += n
    rng = random.Random(5)
    for _ in range(200):
        cmds.append(dict(lits=TEXT[rng.randrange(100000):][:rng.randrange(0, 4)], copy=rng.choice([4, 9, 40, 300, 5000]),
                         dist=("dist", W - rng.randrange(0, 64))))
    cmds.append(dict(lits=b"", copy=1 << 20, dist=("dist", W)))
    cmds.append(dict(lits=b"far", copy=None))
    # a meta-block holds at most 16 MiB: split the command list
    cur, acc, mbs = [], 0, []
    for c in cmds:
        k = len(c.get("lits", b"")) + (c.get("copy") or 0)
        if acc + k > (1 << 24) - 8 and cur:
            cur.append(dict(lits=b"", copy=None)) if False else None
            mbs.append(cur)
            cur, acc = [], 0
        cur.append(c)
        acc += k
    mbs.append(cur)
    for i, m in enumerate(mbs):
        if m[-1].get("copy") is not None:
            m.append(dict(lits=b"|", copy=None))
        blocks.append(("c", dict(cmds=m)))
    res.append(("far_16MiB", dict(wbits=24, blocks=blocks)))
    # a smaller one (1 MiB window, 3 MiB out) that every kernel can be asked to decode
    blocks = [("u", seed_text)]
    size = 65536
    cmds = []
    while size < (1 << 20):
        cmds.append(dict(lits=b"", copy=size, dist=("dist", size)))
        size *= 2
    W20 = (1 << 20) - 16
    for _ in range(300):
        cmds.append(dict(lits=TEXT[rng.randrange(100000):][:rng.randrange(0, 3)], copy=rng.choice([4, 9, 40, 300, 20000]),
                         dist=("dist", W20 - rng.randrange(0, 64))))
    cmds.append(dict(lits=b"end", copy=None))
    blocks.append(("c", dict(cmds=cmds)))
    res.append(("far_1MiB", dict(wbits=20, blocks=blocks)))
    return res


def main():
    if H.reference() is None:
        raise SystemExit("oracle/_ref/libyeast_ref.so is needed (make -C oracle ref)")
    lt = luts()
    blob = bytearray()
    meta = []
    for name, plan, seed in plans():
        stream, want = B.build_stream(plan, dict_word, lt, seed)
        rc, got, used, out = ref_decode(stream)
        assert rc == 0 and out == want and used == len(stream), (name, rc, got, len(want), used, len(stream))
        meta.append(dict(name=name, off=len(blob), len=len(stream), rc=rc, got=got, used=used, **checks(out)))
        blob += stream
        print("%-24s %6d bytes -> %7d" % (name, len(stream), got))
    open(os.path.join(GOLD, "cfg4_streams.bin"), "wb").write(bytes(blob))
    json.dump(dict(generator="tools/brosynth.py + tools/make_cfg4.py", expected="oracle/_ref (the reference decoder)",
                   streams=meta), open(os.path.join(GOLD, "cfg4_streams.json"), "w"), indent=0)
    print("cfg4: %d streams, %d bytes" % (len(meta), len(blob)))

    blob = bytearray()
    meta = []
    for name, plan in far_plans():
        stream, want = B.build_stream(plan, dict_word, lt, 1)
        rc, got, used, out = ref_decode(stream)
        assert rc == 0 and out == want and used == len(stream), (name, rc, got, len(want))
        meta.append(dict(name=name, off=len(blob), len=len(stream), rc=rc, got=got, used=used, **checks(out)))
        blob += stream
        print("%-24s %6d bytes -> %9d" % (name, len(stream), got))
    open(os.path.join(GOLD, "cfg5_streams.bin"), "wb").write(bytes(blob))
    json.dump(dict(generator="tools/brosynth.py + tools/make_cfg4.py", expected="oracle/_ref (the reference decoder)",
                   streams=meta), open(os.path.join(GOLD, "cfg5_streams.json"), "w"), indent=0)
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Randomly assembled brotli streams for the GPU parity tier (tools/brosynth.py builds them, the pinned CPU restatement
says what they decode to): every stream mixes, by a seeded choice, window bits, several meta-blocks (compressed, stored,
metadata), context modes and 64-entry context maps with many trees, NPOSTFIX / NDIRECT, distance context maps, block
switching in the three categories, static dictionary references with transforms, ring-buffer and implicit distances,
overlapping copies and copies longer than one 16-byte piece."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import brosynth as B  # noqa: E402
import make_cfg4 as M  # noqa: E402


def _explicit(cmds):
    """later meta-blocks: only explicit distances (the ring buffer's state is the stream's, not the block's)"""
    out, res = 0, []
    for c in cmds:
        c = dict(c)
        out += len(c.get("lits", b""))
        if c.get("copy") is not None:
            d = c["dist"]
            if d[0] in ("ring", "implicit"):
                d = ("dist", 1 + (out * 7 + c["copy"]) % max(out, 1))
            if d[0] == "dist":
                d = ("dist", max(1, min(d[1], out)))
            c["dist"] = d
            out += c["copy"]
        res.append(c)
    return res


def random_plan(rng):
    wbits = rng.choice([16, 17, 18, 20, 22, 24])
    blocks = []
    nblocks = rng.choice([1, 1, 1, 2, 3])
    first_c = True
    for b in range(nblocks):
        k = rng.random()
        if k < 0.12 and b + 1 < nblocks:
            a = rng.randrange(0, 100000)
            blocks.append(("u", M.TEXT[a:a + rng.choice([1, 17, 300, 2500])]))
            continue
        if k < 0.18 and b + 1 < nblocks:
            blocks.append(("m", b"x" * rng.randrange(0, 40)))
            continue
        nbytes = rng.choice([60, 700, 3000, 9000, 24000])
        cmds = M.text_cmds(rng, nbytes, maxcopy=rng.choice([40, 100, 700]), lit_p=rng.choice([0.15, 0.5, 0.9]))
        # overlapping copies of short periods and long copies
        for _ in range(rng.randrange(0, 6)):
            at = rng.randrange(1, len(cmds))
            cmds.insert(at, dict(lits=b"ab"[:rng.randrange(0, 3)], copy=rng.choice([3, 17, 33, 64, 65, 300]),
                                 dist=("dist", rng.choice([1, 2, 3, 5, 7, 15, 16, 17, 31]))))
        cmds = M.fix_ring(cmds) if first_c else _explicit(cmds)
        # half of the blocks that another block follows end in a copy, not in literals (the next block's first literal
        # contexts are then bytes that a copy wrote: a cooperative-kernel bug of round 3 hid behind blocks that never did)
        if b + 1 < nblocks and len(cmds) > 2 and cmds[-1].get("copy") is None and rng.random() < 0.5:
            cmds = cmds[:-1]
        mb = dict(cmds=cmds)
        if rng.random() < 0.7:
            nL = rng.choice([1, 1, 2, 5]) if rng.random() < 0.5 else 1
            nI = rng.choice([1, 1, 2, 4])
            nD = rng.choice([1, 1, 3])
            if nL * nI * nD > 1:
                def sched(n, unit):
                    seg = [(0, rng.randrange(1, unit))]
                    for _ in range(40):
                        seg.append((rng.randrange(n), rng.choice([1, 1, 2, 5, 17, 40, 100, 400])))
                    return seg
                mb["ntypes"] = dict(L=nL, I=nI, D=nD)
                mb["sched"] = dict(L=sched(nL, 50), I=sched(nI, 20), D=sched(nD, 20))
            trees = rng.choice([1, 2, 7, 23])
            cmap = [rng.randrange(trees) for _ in range(64 * nL)]
            for i in range(trees):
                cmap[i % len(cmap)] = i
            mb.update(modes=[rng.randrange(4) for _ in range(nL)], lit_trees=trees, lit_cmap=cmap)
            dist_trees = rng.choice([1, 2, 4])
            mb.update(dist_trees=dist_trees, dist_cmap=[rng.randrange(dist_trees) for _ in range(4 * nD)])
        if rng.random() < 0.5:
            P = rng.randrange(4)
            mb.update(npostfix=P, ndirect=rng.choice([0, 1, 7, 15]) << P)
        blocks.append(("c", mb))
        first_c = False
    if blocks[-1][0] != "c":
        blocks.append(("c", dict(cmds=[dict(lits=b"end", copy=None)])))
    return dict(wbits=wbits, blocks=blocks)


def dictionary_plan(rng):
    """static dictionary words under random transforms between literal runs (a small window keeps every id reachable)"""
    cmds = []
    for k in range(rng.choice([5, 40, 200])):
        L = rng.randrange(4, 25)
        cmds.append(dict(lits=bytes(rng.choice(b" ,.;\nabc") for _ in range(rng.randrange(0, 4))), copy=L,
                         dist=("dict", L, rng.randrange(1 << B.NDBITS[L]), rng.randrange(121))))
    cmds.append(dict(lits=b"end", copy=None))
    return dict(wbits=rng.choice([10, 11, 12]), blocks=[("c", dict(cmds=cmds, lit_trees=rng.choice([1, 3])))])


def near_plan(rng):
    """copies from just behind the output: distances and lengths around what the batch kernel's LDS ring holds (64 or 128 bytes),
    the limits below which a copy's source is taken from the ring (48, 112) and the chunks output leaves it in (16, 64)"""
    cmds, out = [], 0
    lits = M.TEXT[rng.randrange(0, 50000):][:rng.choice([1, 3, 20, 150])]
    for _ in range(rng.choice([30, 120, 400])):
        d = rng.choice([rng.randrange(1, 34), rng.randrange(30, 70), rng.randrange(90, 135), rng.randrange(1, 300)])
        n = rng.choice([2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 111, 112, 113, 130, rng.randrange(2, 200)])
        out += len(lits)
        cmds.append(dict(lits=lits, copy=n, dist=("dist", max(1, min(d, out)))))
        out += n
        a = rng.randrange(0, 100000)
        lits = M.TEXT[a:a + rng.choice([0, 0, 0, 1, 1, 2, 3, 7, 20])]
    cmds.append(dict(lits=lits + b".", copy=None))
    mb = dict(cmds=cmds)
    if rng.random() < 0.5:
        trees = rng.choice([2, 7])
        cmap = [rng.randrange(trees) for _ in range(64)]
        for i in range(trees):
            cmap[i] = i
        mb.update(modes=[rng.randrange(4)], lit_trees=trees, lit_cmap=cmap)
    return dict(wbits=rng.choice([10, 16, 22]), blocks=[("c", mb)])


def streams(count, seed):
    """[(stream bytes, what the synthesiser says it decodes to)]"""
    rng = random.Random(seed)
    lt = M.luts()
    out = []
    while len(out) < count:
        k = rng.random()
        plan = dictionary_plan(rng) if k < 0.15 else near_plan(rng) if k < 0.30 else random_plan(rng)
        try:
            out.append(B.build_stream(plan, M.dict_word, lt, rng.randrange(1 << 30)))
        except (AssertionError, ValueError, KeyError, IndexError):
            continue  # a combination the synthesiser cannot write (e.g. a distance its code cannot reach)
    return out

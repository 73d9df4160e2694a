#!/usr/bin/env python3
"""brosynth -- a Brotli stream synthesiser (this repo's counterpart of the reference's brogen.cc).

Where brogen.cc turns a tiny command language into exact bit patterns, this module turns a *plan*
(Python dicts: meta-blocks, commands, block-type schedules, context maps, code styles) into a valid
Brotli stream, choosing every format feature explicitly, so that the features the encoder library
never emits can be exercised: all 121 dictionary transforms at every word length, the four literal
context modes with arbitrary context maps, NPOSTFIX/NDIRECT extremes, block switching with type
codes 0 and 1, RLEMAX / inverse-move-to-front context maps, the five simple code shapes,
single-symbol code-length codes, long (15-bit) codes, uncompressed and metadata meta-blocks.

It is a test-fixture generator (tools/make_cfg4.py); nothing in the product imports it.
Format reference: RFC 7932 as implemented by yeast.c (line numbers in oracle/yeast_oracle.c).
"""
import heapq
import random

INS_BASE = [0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594]
INS_EXTRA = [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24]
CPY_BASE = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118]
CPY_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24]
INS_CELL = [0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16]
CPY_CELL = [0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16]
BLEN_BASE = [1, 5, 9, 13, 17, 25, 33, 41, 49, 65, 81, 97, 113, 145, 177, 209, 241, 305, 369, 497, 753, 1265, 2289,
             4337, 8433, 16625]
BLEN_EXTRA = [2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24]
NDBITS = [0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5]
CL_ORDER = [1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15]
# the fixed code for code-length-code lengths: canonical, lengths {0:2, 3:2, 4:2, 2:3, 1:4, 5:4}
CL_VLC = {0: (0b00, 2), 3: (0b01, 2), 4: (0b10, 2), 2: (0b110, 3), 1: (0b1110, 4), 5: (0b1111, 4)}
RING_BACK = [0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1]
RING_DELTA = [0, 0, 0, 0, -1, 1, -2, 2, -3, 3, -1, 1, -2, 2, -3, 3]


class BitWriter:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def put(self, v, nbits):
        assert 0 <= v < (1 << nbits) or nbits == 0, (v, nbits)
        self.acc |= v << self.n
        self.n += nbits

    def code(self, code, length):
        """a prefix code word: its bits go out most significant first"""
        for k in range(length - 1, -1, -1):
            self.put((code >> k) & 1, 1)

    def align(self):
        self.n = (self.n + 7) & ~7

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def limited_lengths(freq, limit):
    """code lengths (<= limit) of a complete prefix code for the symbols of freq (>= 2 of them):
    Huffman, with the frequencies flattened until the depth fits"""
    syms = sorted(freq)
    assert len(syms) >= 2 and len(syms) <= (1 << limit)
    f = {s: max(1, freq[s]) for s in syms}
    while True:
        heap = [(f[s], i, (s,)) for i, s in enumerate(syms)]
        heapq.heapify(heap)
        lens = {s: 0 for s in syms}
        k = len(syms)
        while len(heap) > 1:
            a = heapq.heappop(heap)
            b = heapq.heappop(heap)
            for s in a[2] + b[2]:
                lens[s] += 1
            heapq.heappush(heap, (a[0] + b[0], k, a[2] + b[2]))
            k += 1
        if max(lens.values()) <= limit:
            return lens
        f = {s: (v + 1) // 2 for s, v in f.items()}


def canonical(lens):
    """{sym: (code, len)} for lengths {sym: len > 0}: by length, then by symbol value"""
    out = {}
    code = 0
    last = 0
    for l, s in sorted((l, s) for s, l in lens.items() if l):
        code <<= l - last
        last = l
        out[s] = (code, l)
        code += 1
    assert code == 1 << last, "incomplete code"
    return out


class Code:
    """one prefix code: how it is described in the stream and how symbols are written with it"""

    def __init__(self, num, freq, rng, style=None):
        self.num = num
        style = dict(style or {})
        used = sorted(freq)
        if not used:
            used = [rng.randrange(num) if style.get("random_unused") else 0]
        self.simple = None
        want_simple = style.get("simple", len(used) <= 4 and rng.random() < 0.7)
        if len(used) == 1 or (len(used) <= 4 and want_simple):
            syms = list(used)
            if style.get("pad_simple") and len(syms) < 4:  # unused extra symbols, still a simple code
                pool = [s for s in range(min(num, 64)) if s not in syms]
                rng.shuffle(pool)
                syms += pool[:rng.randrange(0, 5 - len(syms))]
            rng.shuffle(syms)
            shape = len(syms)
            if shape == 4 and (style.get("shape5") or rng.random() < 0.5):
                shape = 5
            # most frequent symbols get the short codes
            syms.sort(key=lambda s: -freq.get(s, 0))
            self.simple = (syms, shape)
            lens_by_pos = {1: [0], 2: [1, 1], 3: [1, 2, 2], 4: [2, 2, 2, 2], 5: [1, 2, 3, 3]}[shape]
            order = list(syms)
            # the decoder sorts symbols inside each equal-length group
            if shape == 2:
                order = sorted(order)
            elif shape == 3:
                order = order[:1] + sorted(order[1:])
            elif shape == 4:
                order = sorted(order)
            elif shape == 5:
                order = order[:2] + sorted(order[2:])
            self.codes = {}
            code = 0
            last = 0
            for s, l in zip(order, lens_by_pos):
                if l == 0:
                    self.codes[s] = (0, 0)
                    continue
                code <<= l - last
                last = l
                self.codes[s] = (code, l)
                code += 1
            return
        f = dict(freq)
        extra = style.get("extra_symbols", 0)  # symbols that are in the code but never used
        pool = [s for s in range(num) if s not in f]
        rng.shuffle(pool)
        for s in pool[:extra]:
            f[s] = 1
        if style.get("skew"):  # geometric frequencies: a deep code (up to the 15-bit limit)
            ss = sorted(f, key=lambda s: -f[s])
            for i, s in enumerate(ss):
                f[s] = 1 << max(0, 24 - i)
        elif style.get("flat"):
            f = {s: 1 for s in f}
        if len(f) < 2:
            f[pool[0] if pool else (used[0] + 1) % num] = 1
        self.lens = limited_lengths(f, 15)
        self.codes = canonical(self.lens)
        self.rle = style.get("rle", rng.random() < 0.6)
        self.hskip_pref = style.get("hskip", rng.choice([0, 2, 3]))

    def write_symbol(self, bw, s):
        c, l = self.codes[s]
        bw.code(c, l)

    def write_descriptor(self, bw):
        num = self.num
        if self.simple is not None:
            syms, shape = self.simple
            abits = 1
            n = 2
            while n < num:
                n <<= 1
                abits += 1
            bw.put(1, 2)
            bw.put(min(shape, 4) - 1, 2)
            for s in syms:
                bw.put(s, abits)
            if shape >= 4:
                bw.put(shape - 4, 1)
            return
        write_lengths(bw, [self.lens.get(s, 0) for s in range(num)], self.rle, self.hskip_pref)


def write_lengths(bw, lens, rle=True, hskip_pref=3):
    """a complex prefix code's descriptor (yeast.c:430-530): the code length code, then the lengths of the symbols
    0 .. last used, with runs as codes 16 and 17 when rle is set; hskip_pref: the largest HSKIP to use when it fits"""
    lens = list(lens)
    while lens and lens[-1] == 0:
        lens.pop()
    # code-length symbols, optionally with the run codes 16 (repeat) and 17 (zeros)
    seq = []  # (symbol, extra value, extra bits)
    if not rle:
        seq = [(l, 0, 0) for l in lens]
    else:
        i = 0
        prev = 8
        while i < len(lens):
            l = lens[i]
            j = i
            while j < len(lens) and lens[j] == l:
                j += 1
            run = j - i
            if l == 0 and run >= 3:
                seq += zero_runs(run)
                i = j
                continue
            if l != 0 and l == prev and run >= 3:
                seq += repeat_runs(run)
                i = j
                continue
            if l != 0 and run >= 4:
                seq.append((l, 0, 0))
                seq += repeat_runs(run - 1)
                prev = l
                i = j
                continue
            seq.append((l, 0, 0))
            if l:
                prev = l
            i += 1
    hist = {}
    for s, _, _ in seq:
        hist[s] = hist.get(s, 0) + 1
    if len(hist) == 1:
        only = next(iter(hist))
        cl_lens = {only: 3}  # single-symbol code-length code: zero bits per symbol (yeast.c:456-459)
        cl_codes = {only: (0, 0)}
    else:
        cl_lens = limited_lengths(hist, 5)
        cl_codes = canonical(cl_lens)
    hskip = 0
    for h in (3, 2):
        if h <= hskip_pref and all(cl_lens.get(CL_ORDER[k], 0) == 0 for k in range(h)):
            hskip = h
            break
    bw.put(hskip, 2)
    space = 32
    for k in range(hskip, 18):
        l = cl_lens.get(CL_ORDER[k], 0)
        c, n = CL_VLC[l]
        bw.code(c, n)
        if l:
            space -= 32 >> l
            if space <= 0:
                break
    for s, ev, eb in seq:
        c, l = cl_codes[s]
        bw.code(c, l)
        if eb:
            bw.put(ev, eb)


def repeat_runs(run):
    """code 16 chains for `run` more copies of the previous non-zero length (yeast.c:480-500):
    each further 16 turns the count so far, r, into (r - 2) * 4 + 3 + extra"""
    return _runs(run, 16, 2)


def zero_runs(run):
    """code 17 chains for a run of `run` zeros (yeast.c:501-515): (r - 2) * 8 + 3 + extra"""
    return _runs(run, 17, 3)


def _runs(run, sym, bits):
    top = (1 << bits) - 1
    # find digits d_k (0..top) with value(d_1..d_m) == run, value = fold(r -> (r-2)<<bits + 3 + d)
    best = None
    for m in range(1, 8):
        lo = hi = 0
        for k in range(m):
            lo = ((lo - 2) << bits if k else 0) + 3
            hi = ((hi - 2) << bits if k else 0) + 3 + top
        if lo <= run <= hi:
            best = m
            break
    if best is None:
        raise ValueError("run too long")
    digits = []
    target = run
    for k in range(best, 0, -1):
        if k == 1:
            d = target - 3
            assert 0 <= d <= top
            digits.append(d)
        else:
            d = (target - 3) & top
            prev = ((target - 3 - d) >> bits) + 2
            digits.append(d)
            target = prev
    digits.reverse()
    # check
    r = 0
    for k, d in enumerate(digits):
        r = ((r - 2) << bits if k else 0) + 3 + d
    assert r == run, (run, digits, r)
    return [(sym, d, bits) for d in digits]


def write_count256(bw, v):
    """block_types(): 1..256"""
    if v == 1:
        bw.put(0, 1)
        return
    bw.put(1, 1)
    c = (v - 1).bit_length() - 1
    bw.put(c, 3)
    bw.put(v - 1 - (1 << c), c)


def blen_symbol(count):
    for k in range(25, -1, -1):
        if count >= BLEN_BASE[k]:
            assert count - BLEN_BASE[k] < (1 << BLEN_EXTRA[k])
            return k, count - BLEN_BASE[k], BLEN_EXTRA[k]
    raise ValueError(count)


def len_code(v, base, extra):
    for k in range(23, -1, -1):
        if v >= base[k]:
            assert v - base[k] < (1 << extra[k]), v
            return k
    raise ValueError(v)


def context_id(p1, p2, mode, luts):
    if mode == 0:
        return p1 & 63
    if mode == 1:
        return p1 >> 2
    if mode == 2:
        return luts[0][p1] | luts[1][p2]
    return (luts[2][p1] << 3) | luts[2][p2]


def write_context_map(bw, cmap, trees, rng, style):
    """context_map(): yeast.c:589-647"""
    rlemax = style.get("rlemax", rng.choice([0, 0, 1, 2, 4, 6, 16]))
    imtf = style.get("imtf", rng.random() < 0.5)
    vals = list(cmap)
    if imtf:  # forward move-to-front, so that the decoder's inverse gives cmap back
        tab = list(range(256))
        out = []
        for v in vals:
            i = tab.index(v)
            out.append(i)
            tab.pop(i)
            tab.insert(0, v)
        vals = out
    seq = []
    i = 0
    while i < len(vals):
        if vals[i] == 0 and rlemax:
            j = i
            while j < len(vals) and vals[j] == 0:
                j += 1
            run = j - i
            while run:
                k = min(rlemax, run.bit_length() - 1)
                if k == 0:
                    seq.append((0, 0, 0))
                    run -= 1
                    continue
                take = min(run, (1 << (k + 1)) - 1)
                seq.append((k, take - (1 << k), k))
                run -= take
            i = j
        else:
            seq.append((vals[i] + rlemax if vals[i] else 0, 0, 0))
            i += 1
    hist = {}
    for s, _, _ in seq:
        hist[s] = hist.get(s, 0) + 1
    code = Code(rlemax + trees, hist, rng, style.get("code"))
    if rlemax:
        bw.put(1, 1)
        bw.put(rlemax - 1, 4)
    else:
        bw.put(0, 1)
    code.write_descriptor(bw)
    for s, ev, eb in seq:
        code.write_symbol(bw, s)
        if eb:
            bw.put(ev, eb)
    bw.put(1 if imtf else 0, 1)


class Schedule:
    """block types of one category: [(type, count), ...]; the first type must be 0"""

    def __init__(self, num, segments, rng, prefer_short=0.7):
        self.num = num
        self.seg = list(segments) if num > 1 else []
        self.i = 0
        self.left = self.seg[0][1] if self.seg else None
        self.type = 0
        self.prev = 1
        self.rng = rng
        self.prefer_short = prefer_short
        if self.seg:
            assert self.seg[0][0] == 0

    def step(self, tokens, cat):
        """called before every symbol of the category; appends a block-switch command if due"""
        if self.num == 1:
            return 0
        if self.left == 0:
            self.i += 1
            if self.i >= len(self.seg):  # keep going with long blocks of a random type
                self.seg.append((self.rng.randrange(self.num), 1 << 20))
            t, cnt = self.seg[self.i]
            choices = [t + 2]
            if t == self.prev:
                choices.append(0)
            if t == (self.type + 1) % self.num:
                choices.append(1)
            sym = t + 2
            if len(choices) > 1 and self.rng.random() < self.prefer_short:
                sym = self.rng.choice(choices[1:])
            tokens.append(("sym", ("BT", cat), sym))
            k, ev, eb = blen_symbol(cnt)
            tokens.append(("sym", ("BC", cat), k))
            tokens.append(("bits", ev, eb))
            self.prev = self.type
            self.type = t
            self.left = cnt
        self.left -= 1
        return self.type


class Model:
    """decoder state mirrored while a stream is planned: output so far, distance ring"""

    def __init__(self, wbits, dict_word, luts):
        self.wbits = wbits
        self.wsize = (1 << wbits) - 16
        self.out = bytearray()
        self.ring = [16, 15, 11, 4]
        self.ring_ptr = 3
        self.dict_word = dict_word
        self.luts = luts

    def max_dist(self):
        return min(len(self.out), self.wsize)

    def ring_value(self, code):
        return self.ring[(self.ring_ptr - RING_BACK[code]) & 3] + RING_DELTA[code]

    def push(self, d):
        self.ring_ptr = (self.ring_ptr + 1) & 3
        self.ring[self.ring_ptr] = d


def write_wbits(bw, wbits):
    if wbits == 16:
        bw.put(0, 1)
    elif wbits >= 18:
        bw.put(1, 1)
        bw.put(wbits - 17, 3)
    elif wbits == 17:
        bw.put(1, 1)
        bw.put(0, 3)
        bw.put(0, 3)
    else:
        bw.put(1, 1)
        bw.put(0, 3)
        bw.put(wbits - 8, 3)


def distance_symbol(dist, npostfix, ndirect):
    """(symbol, extra value, extra bits) for a distance that is not taken from the ring"""
    if dist <= ndirect:
        return 15 + dist, 0, 0
    v = dist - ndirect - 1
    lo = v & ((1 << npostfix) - 1)
    t = (v >> npostfix) + 4
    x = t.bit_length() - 2
    b = (t >> x) & 1
    extra = t & ((1 << x) - 1)
    n = ((x - 1) << (npostfix + 1)) | (b << npostfix) | lo
    assert x <= 24
    return n + ndirect + 16, extra, x


def compressed_metablock(bw, model, mb, rng, last):
    """One compressed meta-block.  mb: dict with
         cmds:     [{'lits': bytes, 'copy': n | None, 'dist': ('ring', code) | ('dist', d) | ('dict', L, idx, t) |
                     ('implicit',)}]   ('copy' None only in the last command: the meta-block ends after its literals)
         npostfix, ndirect (ndirect already shifted: a multiple of 1 << npostfix, < 16 << npostfix)
         ntypes: {'L': n, 'I': n, 'D': n}; sched: {'L': [(type, count)...], ...}
         modes: [mode per literal block type]; lit_trees, lit_cmap (64 * ntypes L); dist_trees, dist_cmap (4 * ntypes D)
         style: {'lit': {...}, 'iac': {...}, 'dist': {...}, 'bt': {...}, 'bc': {...}, 'cmapL': {...}, 'cmapD': {...}}"""
    P, ND = mb.get("npostfix", 0), mb.get("ndirect", 0)
    nt = mb.get("ntypes", {"L": 1, "I": 1, "D": 1})
    sched = {c: Schedule(nt[c], mb.get("sched", {}).get(c, [(0, 1 << 20)]), rng) for c in "LID"}
    first_count = {c: (sched[c].seg[0][1] if nt[c] > 1 else None) for c in "LID"}
    modes = mb.get("modes", [0] * nt["L"])
    lit_trees = mb.get("lit_trees", 1)
    lit_cmap = mb.get("lit_cmap", [0] * (64 * nt["L"]))
    dist_trees = mb.get("dist_trees", 1)
    dist_cmap = mb.get("dist_cmap", [0] * (4 * nt["D"]))
    style = mb.get("style", {})
    tokens = []
    start = len(model.out)
    ncmd = len(mb["cmds"])
    for ci, cmd in enumerate(mb["cmds"]):
        lits = cmd.get("lits", b"")
        copy = cmd.get("copy")
        dist = cmd.get("dist")
        final = copy is None
        assert not final or ci == ncmd - 1
        enc_copy = 2 if final else copy  # a copy length must be coded even when it is never used
        ic = len_code(len(lits), INS_BASE, INS_EXTRA)
        cc = len_code(enc_copy, CPY_BASE, CPY_EXTRA)
        implicit = (not final) and dist[0] == "implicit"
        cell = None
        for k in range(11):
            if INS_CELL[k] <= ic < INS_CELL[k] + 8 and CPY_CELL[k] <= cc < CPY_CELL[k] + 8 and (k < 2) == implicit:
                cell = k
                break
        if cell is None:  # an implicit distance is not available for these lengths: spell the ring code out
            assert implicit
            implicit = False
            dist = ("ring", 0)
            for k in range(2, 11):
                if INS_CELL[k] <= ic < INS_CELL[k] + 8 and CPY_CELL[k] <= cc < CPY_CELL[k] + 8:
                    cell = k
                    break
        sym = cell * 64 + ((ic - INS_CELL[cell]) << 3) + (cc - CPY_CELL[cell])
        bt = sched["I"].step(tokens, "I")
        tokens.append(("sym", ("I", bt), sym))
        tokens.append(("bits", len(lits) - INS_BASE[ic], INS_EXTRA[ic]))
        tokens.append(("bits", enc_copy - CPY_BASE[cc], CPY_EXTRA[cc]))
        for b in lits:
            bt = sched["L"].step(tokens, "L")
            p1 = model.out[-1] if len(model.out) >= 1 else 0
            p2 = model.out[-2] if len(model.out) >= 2 else 0
            tree = lit_cmap[bt * 64 + context_id(p1, p2, modes[bt], model.luts)]
            tokens.append(("sym", ("L", tree), b))
            model.out.append(b)
        if final:
            break
        mx = model.max_dist()
        if implicit:
            d = model.ring[model.ring_ptr]
        else:
            if dist[0] == "ring":
                d = model.ring_value(dist[1])
                dsym, ev, eb = dist[1], 0, 0
            elif dist[0] == "dict":
                _, L, idx, t = dist
                assert L == copy
                d = mx + 1 + (t << NDBITS[L]) + idx
                dsym, ev, eb = distance_symbol(d, P, ND)
            else:
                d = dist[1]
                dsym, ev, eb = distance_symbol(d, P, ND)
            bt = sched["D"].step(tokens, "D")
            tree = dist_cmap[bt * 4 + (3 if copy > 4 else copy - 2)]
            tokens.append(("sym", ("D", tree), dsym))
            tokens.append(("bits", ev, eb))
            if dsym != 0 and 0 < d <= mx:
                model.push(d)
        assert d > 0, "planned distance is not positive"
        if d > mx:
            word = model.dict_word(copy, d - mx - 1)
            assert word is not None, ("invalid dictionary reference planned", copy, d - mx - 1)
            model.out += word
        else:
            for _ in range(copy):
                model.out.append(model.out[-d])
    mlen = len(model.out) - start
    assert 1 <= mlen <= (1 << 24)

    # histograms -> codes
    hist = {}
    for t in tokens:
        if t[0] == "sym":
            hist.setdefault(t[1], {})
            hist[t[1]][t[2]] = hist[t[1]].get(t[2], 0) + 1
    ndist = 16 + ND + (48 << P)
    codes = {}
    for tree in range(lit_trees):
        codes[("L", tree)] = Code(256, hist.get(("L", tree), {}), rng, style.get("lit"))
    for bt in range(nt["I"]):
        codes[("I", bt)] = Code(704, hist.get(("I", bt), {}), rng, style.get("iac"))
    for tree in range(dist_trees):
        codes[("D", tree)] = Code(ndist, hist.get(("D", tree), {}), rng, style.get("dist"))
    for c in "LID":
        if nt[c] > 1:
            codes[("BT", c)] = Code(nt[c] + 2, hist.get(("BT", c), {}), rng, style.get("bt"))
            h = dict(hist.get(("BC", c), {}))
            k0 = blen_symbol(first_count[c])[0]
            h[k0] = h.get(k0, 0) + 1
            codes[("BC", c)] = Code(26, h, rng, style.get("bc"))

    # header
    bw.put(1 if last else 0, 1)
    if last:
        bw.put(0, 1)  # not empty
    nib = 4
    while (mlen - 1) >> (4 * nib):
        nib += 1
    if mb.get("nibbles"):
        nib = max(nib, mb["nibbles"]) if (mlen - 1) >> (4 * (mb["nibbles"] - 1)) else nib
    bw.put(nib - 4, 2)
    bw.put(mlen - 1, 4 * nib)
    if not last:
        bw.put(0, 1)  # compressed
    for c in "LID":
        write_count256(bw, nt[c])
        if nt[c] > 1:
            codes[("BT", c)].write_descriptor(bw)
            codes[("BC", c)].write_descriptor(bw)
            k, ev, eb = blen_symbol(first_count[c])
            codes[("BC", c)].write_symbol(bw, k)
            bw.put(ev, eb)
    bw.put(P, 2)
    bw.put(ND >> P, 4)
    for m in modes:
        bw.put(m, 2)
    write_count256(bw, lit_trees)
    if lit_trees > 1:
        write_context_map(bw, lit_cmap, lit_trees, rng, style.get("cmapL", {}))
    write_count256(bw, dist_trees)
    if dist_trees > 1:
        write_context_map(bw, dist_cmap, dist_trees, rng, style.get("cmapD", {}))
    for tree in range(lit_trees):
        codes[("L", tree)].write_descriptor(bw)
    for bt in range(nt["I"]):
        codes[("I", bt)].write_descriptor(bw)
    for tree in range(dist_trees):
        codes[("D", tree)].write_descriptor(bw)
    for t in tokens:
        if t[0] == "sym":
            codes[t[1]].write_symbol(bw, t[2])
        elif t[2]:
            bw.put(t[1], t[2])


def uncompressed_metablock(bw, model, data):
    mlen = len(data)
    assert 1 <= mlen <= (1 << 24)
    bw.put(0, 1)
    nib = 4
    while (mlen - 1) >> (4 * nib):
        nib += 1
    bw.put(nib - 4, 2)
    bw.put(mlen - 1, 4 * nib)
    bw.put(1, 1)
    bw.align()
    for b in data:
        bw.put(b, 8)
    model.out += data


def metadata_metablock(bw, payload, last=False):
    bw.put(1 if last else 0, 1)
    if last:
        bw.put(0, 1)
    bw.put(3, 2)
    bw.put(0, 1)
    n = len(payload)
    nbytes = 0 if n == 0 else ((n - 1).bit_length() + 7) // 8 or 1
    bw.put(nbytes, 2)
    if nbytes:
        bw.put(n - 1, 8 * nbytes)
    bw.align()
    for b in payload:
        bw.put(b, 8)


def build_stream(plan, dict_word, luts, seed=0):
    """plan: {'wbits': w, 'blocks': [('c', mb) | ('u', bytes) | ('m', bytes) | ('e',)]} -> (stream, output)"""
    rng = random.Random(seed)
    bw = BitWriter()
    write_wbits(bw, plan["wbits"])
    model = Model(plan["wbits"], dict_word, luts)
    blocks = plan["blocks"]
    for i, blk in enumerate(blocks):
        is_last = i == len(blocks) - 1
        if blk[0] == "c":
            compressed_metablock(bw, model, blk[1], rng, is_last)
        elif blk[0] == "u":
            assert not is_last
            uncompressed_metablock(bw, model, blk[1])
        elif blk[0] == "m":
            metadata_metablock(bw, blk[1], is_last)
        elif blk[0] == "e":
            assert is_last
            bw.put(1, 1)
            bw.put(1, 1)
    return bw.bytes(), bytes(model.out)

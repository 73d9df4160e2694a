#!/usr/bin/env python3
"""brogen.py -- this repo's counterpart of the reference's test-stream generator.

Reads the command language of brogen.cc (commands brogen.cc:44-61, help text :64-83) and writes the
exact bit pattern it describes, so that the reference's good/*.gen and bad/*.gen files -- and new traps
written in the same notation -- can be turned into streams here (the reference's program does not build
in this image: it needs BSD digittoint()).

  b n x      the low n bits of x                      bound x   fill to a byte boundary with x
  w n        WBITS header for n window bits           last n    the next meta-block is (not) the last
  m n        compressed meta-block lead-in, n bytes   u n       uncompressed meta-block lead-in
  e n        empty meta-block, n bytes of metadata (-1: last and empty)
  lit x ...  bytes (numbers and "strings"), from a byte boundary
  types n    coded number of block types              len n     one code length code length (0..5)
  s id t a s ...   simple prefix code of type t over a-bit symbols
  c id b s b s ... complex prefix code: lengths b for symbols s
  p id s ...       symbols coded with code id
  ;  ends a command    #  starts a comment

Pinning: tests/test_oracle_golden.py runs every .gen file of the reference through this program and
compares with the bytes the reference's own tests produce from them (SURVEY section 4's table, which is
tests/golden/crafted.json).  No .gen file of the reference uses `c`, so the bytes written for a complex
code are unpinned; the test checks instead that the decoder restatement reads back the code that was asked for.
Usage: tools/brogen.py < x.gen > x.compressed, or brogen(text) -> bytes.
"""
import re
import sys


class Bits:
    """LSB-first bit writer (bout(): brogen.cc:221-255)."""

    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, n, val):
        self.acc |= (val & ((1 << n) - 1)) << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def flush(self, fill=0):
        if self.n:
            self.acc |= fill << self.n
            self.out.append(self.acc & 255)
            self.acc = 0
            self.n = 0


def canonical(count, symbols):
    """{symbol: (bits, code reversed for the stream)} from counts per length and the symbols in canonical
    order (encode(): brogen.cc:274-291)."""
    enc = {}
    count = list(count)
    length, code, k = 0, 0, 0
    while True:
        while count[length] == 0:
            length += 1
        enc[symbols[k]] = (length, code)
        k += 1
        count[length] -= 1
        bit = 1 << length  # add one to the code from its far end
        while True:
            bit >>= 1
            if not bit:
                break
            code ^= bit
            if code & bit:
                break
        if code == 0:
            return enc


def huffman_lengths(freq):
    """Code lengths for frequencies sorted ascending (huffman(): huff.c:14-70, Moffat and Katajainen's
    in-place method); the lengths come out longest first."""
    n = len(freq)
    if n == 0:
        return []
    if n == 1:
        return [0]
    a = list(freq)
    a[0] += a[1]
    root, leaf = 0, 2
    for nxt in range(1, n - 1):
        if leaf >= n or a[root] < a[leaf]:
            a[nxt] = a[root]
            a[root] = nxt
            root += 1
        else:
            a[nxt] = a[leaf]
            leaf += 1
        if leaf >= n or (root < nxt and a[root] < a[leaf]):
            a[nxt] += a[root]
            a[root] = nxt
            root += 1
        else:
            a[nxt] += a[leaf]
            leaf += 1
    a[n - 2] = 0
    for nxt in range(n - 3, -1, -1):
        a[nxt] = a[a[nxt]] + 1
    avail, used, depth = 1, 0, 0
    root = nxt = n - 1
    while avail:
        while root and a[root - 1] == depth:
            used += 1
            root -= 1
        while avail > used:
            a[nxt] = depth
            nxt -= 1
            avail -= 1
        avail = used << 1
        depth += 1
        used = 0
    return a


def flatten(bits, limit):
    """Limit code lengths (sorted longest first) to `limit` bits keeping the code complete
    (flatten(): flatten.c:5-63).  Returns False if that cannot be done."""
    n = len(bits)
    if n == 0 or bits[0] <= limit:
        return True
    if (1 << limit) < n:
        return False
    mx = bits[0]
    k, debt, to = 0, 0, 1 << (mx - limit)
    while True:
        debt += to - (1 << (mx - bits[k]))
        bits[k] = limit
        k += 1
        if not (k < n and bits[k] > limit):
            break
    debt >>= mx - limit
    while k < n and bits[k] == limit:
        k += 1
    while k < n:
        pay = (1 << (limit - bits[k])) - 1
        bits[k] = limit
        k += 1
        debt -= pay
        if debt <= 0:
            break
    if debt == 0:
        return True
    k = n
    while k and bits[k - 1] == 1:
        k -= 1
    while k:
        spend = 1 << (limit - bits[k - 1])
        if debt + spend > 0:
            k -= 1
            continue
        k -= 1
        bits[k] -= 1
        debt += spend
        if debt == 0:
            break
    return debt == 0


def write_simple(w, kind, bits, syms):
    """simple(): brogen.cc:298-338."""
    num = 4 if kind == 5 else kind
    w.put(2, 1)
    w.put(2, num - 1)
    for s in syms[:num]:
        w.put(bits, s)
    if num >= 4:
        w.put(1, kind - 4)
    s = list(syms)
    if kind == 1:
        return canonical([1, 0, 0, 0], s)
    if kind == 2:
        return canonical([0, 2, 0, 0], sorted(s))
    if kind == 3:
        return canonical([0, 1, 2, 0], [s[0]] + sorted(s[1:3]))
    if kind == 4:
        return canonical([0, 0, 4, 0], sorted(s))
    return canonical([0, 1, 1, 2], s[:2] + sorted(s[2:4]))


def write_complex(w, desc):
    """complex(): brogen.cc:346-518.  desc: [(length, symbol)], a complete code."""
    desc = sorted(desc, key=lambda d: d[1])
    inst = []  # (instruction 0..17, extra bits value)
    state = {"rep": 0, "len": 0, "last": 8}

    def emit():
        while state["rep"]:
            if state["rep"] < 3 or state["len"] != state["last"]:
                inst.append((state["len"], 0))
                state["last"] = state["len"]
                state["rep"] -= 1
            if state["rep"] >= 3:
                dig = []
                state["rep"] -= 2
                while True:
                    state["rep"] -= 1
                    dig.append(state["rep"] & 3)
                    state["rep"] >>= 2
                    if not state["rep"]:
                        break
                for d in reversed(dig):
                    inst.append((16, d))
    nxt = 0
    for length, sym in desc:
        if nxt < sym:
            emit()
            zeros = sym - nxt
            if zeros < 3:
                inst.extend([(0, 0)] * zeros)
            else:
                dig = []
                zeros -= 2
                while True:
                    zeros -= 1
                    dig.append(zeros & 7)
                    zeros >>= 3
                    if not zeros:
                        break
                for d in reversed(dig):
                    inst.append((17, d))
            nxt = sym
        if state["rep"] and state["len"] != length:
            emit()
        state["len"] = length
        state["rep"] += 1
        nxt += 1
    emit()
    freq = [0] * 18
    for i, _ in inst:
        freq[i] += 1
    idesc = [[freq[i], i] for i in range(18) if freq[i]]  # [frequency -> length, instruction]
    if len(idesc) > 1:
        idesc.sort()
        lens = huffman_lengths([d[0] for d in idesc])
        assert flatten(lens, 5)
        for d, l in zip(idesc, lens):
            d[0] = l
        idesc.sort()
        count = [0] * 6
        for l, _ in idesc:
            count[l] += 1
        icode = canonical(count, [i for _, i in idesc])
    else:
        idesc[0][0] = 3
        icode = {idesc[0][1]: (0, 0)}
    order = [4, 0, 1, 2, 3, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 8, 6]
    lst = [0] * 18
    for l, i in idesc:
        lst[order[i]] = l
    start = 0 if (lst[0] or lst[1]) else 2 if lst[2] else 3
    w.put(2, start)
    end = 17
    if len(idesc) > 1:
        while end and lst[end] == 0:
            end -= 1
    lencode = canonical([0, 0, 3, 1, 2], [0, 3, 4, 2, 1, 5])
    for k in range(start, end + 1):
        w.put(*lencode[lst[k]])
    for i, x in inst:
        w.put(*icode[i])
        if i > 15:
            w.put(i - 14, x)
    desc = sorted(desc)
    count = [0] * 16
    for l, _ in desc:
        count[l] += 1
    return canonical(count, [s for _, s in desc])


_ESC = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11}


def _tokens(text):
    """Commands with their literal arguments: [(name, [values])] (main(): brogen.cc:531-566; getlit(): :138-193)."""
    pos, n = 0, len(text)
    out = []
    cur = None
    while pos < n:
        ch = text[pos]
        if ch.isspace():
            pos += 1
        elif ch == "#":
            while pos < n and text[pos] != "\n":
                pos += 1
        elif ch == ";":
            cur = None
            pos += 1
        elif ch == '"' and cur is not None:
            pos += 1
            while pos < n and text[pos] != '"':
                c = text[pos]
                pos += 1
                if c == "\\" and pos < n:
                    c = text[pos]
                    pos += 1
                    if c in _ESC:
                        cur[1].append(_ESC[c])
                    elif c == "x":
                        m = re.match(r"[0-9a-fA-F]*", text[pos:])
                        pos += m.end()
                        cur[1].append(int(m.group() or "0", 16) & 255)
                    elif c in "01234567":
                        m = re.match(r"[0-7]{0,2}", text[pos:])
                        pos += m.end()
                        cur[1].append(int(c + m.group(), 8) & 255)
                    else:
                        cur[1].append(ord(c))
                else:
                    cur[1].append(ord(c))
            pos += 1
        else:
            m = re.match(r"[^\s;#]+", text[pos:])
            tok = m.group()
            pos += m.end()
            if cur is not None and (tok[0].isdigit() or tok[0] in "+-"):
                try:
                    cur[1].append(int(tok, 0) if not re.fullmatch(r"[+-]?0[0-7]+", tok) else int(tok, 8))
                except ValueError:
                    sys.stderr.write("! invalid number %s (ignored)\n" % tok)
            else:
                cur = (tok.lower(), [])
                out.append(cur)
    return out


def brogen(text):
    w = Bits()
    codes = {}
    last = 0

    def parm(vals, default, low, high, name):
        v = vals.pop(0) if vals else default
        if v < low or v > high:
            sys.stderr.write("! invalid %s %d\n" % (name, v))
            return None
        return v
    for cmd, vals in _tokens(text):
        if cmd == "b":
            p = parm(vals, 1, 0, 63, "bits count")
            q = parm(vals, 0, 0, (1 << (p or 0)) - 1, "bits value")
            if p is not None and q is not None:
                w.put(p, q)
        elif cmd == "bound":
            p = parm(vals, 0, 0, 127, "bound fill bits")
            w.flush(p or 0)
        elif cmd == "w":
            p = parm(vals, 16, 10, 24, "wbits")
            if p is not None:
                w.put(1, 0 if p == 16 else 1)
                if p != 16:
                    w.put(3, 0 if p < 18 else p - 17)
                    if p < 18:
                        w.put(3, 0 if p == 17 else p - 8)
        elif cmd == "last":
            v = parm(vals, 1, 0, 1, "last")
            last = last if v is None else v
        elif cmd in ("m", "u"):
            p = parm(vals, 1, 1, 1 << 24, "meta-block length")
            if p is None:
                continue
            if cmd == "u" and last:
                sys.stderr.write("last block cannot be uncompressed\n")
                continue
            if cmd == "m" and last:
                w.put(2, 1)
            else:
                w.put(1, 0)
            q = 6 if p > (1 << 20) else 5 if p > (1 << 16) else 4
            w.put(2, q - 4)
            w.put(q << 2, p - 1)
            if cmd == "u":
                w.put(1, 1)
            elif not last:
                w.put(1, 0)
        elif cmd == "e":
            p = parm(vals, 0, -1, 1 << 24, "meta-data length")
            if p is None:
                continue
            if last or p == -1:
                if p == -1:
                    w.put(2, 3)
                    continue
                w.put(2, 1)
            else:
                w.put(1, 0)
            w.put(2, 3)
            w.put(1, 0)
            q = 0 if p <= 0 else 1 if p <= (1 << 8) else 2 if p <= (1 << 16) else 3
            w.put(2, q)
            w.put(q << 3, p - 1)
        elif cmd == "lit":
            w.flush()
            w.out.extend(v & 255 for v in vals)
            vals.clear()
        elif cmd == "types":
            p = parm(vals, 1, 1, 256, "number of block types")
            if p is None:
                continue
            w.put(1, 1 if p > 1 else 0)
            if p > 1:
                q = 0
                while (1 << (q + 1)) < p:
                    q += 1
                w.put(3, q)
                if q:
                    w.put(q, p - 1 - (1 << q))
        elif cmd == "len":
            p = parm(vals, 0, 0, 5, "code length")
            if p is not None:
                w.put(*[(2, 0), (4, 7), (3, 3), (2, 2), (2, 1), (4, 15)][p])
        elif cmd == "s":
            ident = parm(vals, 0, -(1 << 63), 1 << 63, "id")
            kind = parm(vals, 0, 1, 5, "simple code type")
            bits = parm(vals, 0, 1, 10, "alphabet bits") if kind is not None else None
            if bits is None or len(vals) != (4 if kind == 5 else kind):
                sys.stderr.write("invalid parameters for s -- skipping\n")
                vals.clear()
                continue
            limit = min(1 << bits, 704)
            if any(v < 0 or v >= limit for v in vals) or len(set(vals)) != len(vals):
                sys.stderr.write("invalid symbol values -- skipping\n")
                vals.clear()
                continue
            codes[ident] = write_simple(w, kind, bits, list(vals))
            vals.clear()
        elif cmd == "c":
            ident = parm(vals, 0, -(1 << 63), 1 << 63, "id")
            if ident is None or len(vals) & 1:
                sys.stderr.write("invalid code id or missing symbol -- skipping\n")
                vals.clear()
                continue
            desc = [(vals[k], vals[k + 1]) for k in range(0, len(vals), 2)]
            vals.clear()
            syms = [s for _, s in desc]
            if any(s < 0 or s >= 704 for s in syms) or len(set(syms)) != len(syms) or \
                    any(l < 0 or l > 15 for l, _ in desc):
                sys.stderr.write("invalid length or symbol values -- skipping\n")
                continue
            if sum(1 << (15 - l) for l, _ in desc) != 1 << 15:
                sys.stderr.write("incomplete code -- skipping\n")
                continue
            codes[ident] = write_complex(w, desc)
        elif cmd == "p":
            ident = parm(vals, 0, -(1 << 63), 1 << 63, "id")
            if ident in codes:
                for s in vals:
                    if s in codes[ident]:
                        w.put(*codes[ident][s])
                    else:
                        sys.stderr.write("symbol %d not found in code %d\n" % (s, ident))
            else:
                sys.stderr.write("code %s not found\n" % ident)
            vals.clear()
        elif cmd == "help":
            sys.stderr.write(__doc__)
        else:
            sys.stderr.write("! unknown command: %s\n" % cmd)
        if vals:
            sys.stderr.write("%d extraneous parameters for %s ignored\n" % (len(vals), cmd))
    w.flush()
    return bytes(w.out)


if __name__ == "__main__":
    sys.stdout.buffer.write(brogen(sys.stdin.read()))

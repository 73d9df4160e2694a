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

The command language is the reference's; the construction of the codes is this repo's own (tools/brosynth.py:
heap Huffman with frequency flattening for the length limit, canonical assignment by (length, symbol), its own
planner for the run-length symbols 16 and 17), not the reference's huff.c / flatten.c / complex().

Pinning: tests/test_oracle_golden.py runs every .gen file of the reference through this program and
compares with the bytes the reference's own tests produce from them (SURVEY section 4's table, which is
tests/golden/crafted.json).  No .gen file of the reference uses `c`, so the bytes written for a complex
code are unpinned; the test checks instead that the decoder restatement reads back the code that was asked for.
Usage: tools/brogen.py < x.gen > x.compressed, or brogen(text) -> bytes.
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from brosynth import BitWriter, canonical, write_lengths  # this repo's own code construction and descriptor writer

# the variable-length code of the code length code lengths 0..5 (RFC 7932 section 3.5), as (bits, value as the stream holds it)
LEN_BITS = [(2, 0), (4, 7), (3, 3), (2, 2), (2, 1), (4, 15)]


class Out(BitWriter):
    """The stream under construction: brosynth's bit writer behind the argument order of this file's commands."""

    def bits(self, n, val):
        self.put(val & ((1 << n) - 1), n)

    def word(self, cw):
        """a code word (code, length), most significant bit first"""
        self.code(*cw)

    def to_boundary(self, fill=0):
        pad = -self.n & 7
        self.put(fill & ((1 << pad) - 1), pad)

    def raw(self, data):
        self.to_boundary()
        for v in data:
            self.put(v & 255, 8)


def simple_code(w, kind, abits, syms):
    """`s`: a simple prefix code (RFC 7932 section 3.4; what brogen.cc:298-338 writes): NSYM - 1, the symbols,
    the tree-select bit for four symbols.  Returns {symbol: (code, length)}: the lengths are fixed by the kind,
    and within one length the decoder takes the symbols in rising order."""
    num = 4 if kind == 5 else kind
    w.bits(2, 1)
    w.bits(2, num - 1)
    for v in syms[:num]:
        w.bits(abits, v)
    if num == 4:
        w.bits(1, kind - 4)
    shape = {1: [0], 2: [1, 1], 3: [1, 2, 2], 4: [2, 2, 2, 2], 5: [1, 2, 3, 3]}[kind]
    if kind == 1:
        return {syms[0]: (0, 0)}
    return canonical(dict(zip(syms, shape)))


def complex_code(w, desc):
    """`c`: a complex prefix code from [(length, symbol)] (a complete code; what brogen.cc:346-518 writes).  The
    descriptor is laid out by brosynth.write_lengths: run-length symbols where a run is long enough, a
    length-limited Huffman code over the symbols of that sequence.  Returns {symbol: (code, length)}."""
    lens = [0] * (max(sym for _, sym in desc) + 1)
    for length, sym in desc:
        lens[sym] = length
    write_lengths(w, lens, rle=True, hskip_pref=3)
    return canonical({sym: length for length, sym in desc if length})


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
    w = Out()
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
                w.bits(p, q)
        elif cmd == "bound":
            p = parm(vals, 0, 0, 127, "bound fill bits")
            w.to_boundary(p or 0)
        elif cmd == "w":
            p = parm(vals, 16, 10, 24, "wbits")
            if p is not None:
                w.bits(1, 0 if p == 16 else 1)
                if p != 16:
                    w.bits(3, 0 if p < 18 else p - 17)
                    if p < 18:
                        w.bits(3, 0 if p == 17 else p - 8)
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
                w.bits(2, 1)
            else:
                w.bits(1, 0)
            q = 6 if p > (1 << 20) else 5 if p > (1 << 16) else 4
            w.bits(2, q - 4)
            w.bits(q << 2, p - 1)
            if cmd == "u":
                w.bits(1, 1)
            elif not last:
                w.bits(1, 0)
        elif cmd == "e":
            p = parm(vals, 0, -1, 1 << 24, "meta-data length")
            if p is None:
                continue
            if last or p == -1:
                if p == -1:
                    w.bits(2, 3)
                    continue
                w.bits(2, 1)
            else:
                w.bits(1, 0)
            w.bits(2, 3)
            w.bits(1, 0)
            q = 0 if p <= 0 else 1 if p <= (1 << 8) else 2 if p <= (1 << 16) else 3
            w.bits(2, q)
            w.bits(q << 3, p - 1)
        elif cmd == "lit":
            w.raw(vals)
            vals.clear()
        elif cmd == "types":
            p = parm(vals, 1, 1, 256, "number of block types")
            if p is None:
                continue
            w.bits(1, 1 if p > 1 else 0)
            if p > 1:
                q = 0
                while (1 << (q + 1)) < p:
                    q += 1
                w.bits(3, q)
                if q:
                    w.bits(q, p - 1 - (1 << q))
        elif cmd == "len":
            p = parm(vals, 0, 0, 5, "code length")
            if p is not None:
                w.bits(*LEN_BITS[p])
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
            codes[ident] = simple_code(w, kind, bits, list(vals))
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
            codes[ident] = complex_code(w, desc)
        elif cmd == "p":
            ident = parm(vals, 0, -(1 << 63), 1 << 63, "id")
            if ident in codes:
                for s in vals:
                    if s in codes[ident]:
                        w.word(codes[ident][s])
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
    w.to_boundary()
    return w.bytes()


if __name__ == "__main__":
    sys.stdout.buffer.write(brogen(sys.stdin.read()))

#!/usr/bin/env python3
"""Regenerate the static spec data the decoder needs (run in the build container only).

Reads, as text/data, two files of the reference tree and re-emits them in this
repo's own layouts:

  * /root/reference/brotli-dict  (122,784-byte RFC 7932 Appendix A dictionary, data file;
    sha256 in /root/reference/brotli-dict.sha256) -> brotli_amd/data/brotli_dictionary.bin
  * /root/reference/xforms.h:1-123 (the 121 RFC 7932 Appendix B transforms)
    -> brotli_amd/data/transforms.bin: 121 fixed 20-byte records
         byte 0       : op   (0 identity, 1 uppercase-first, 2 uppercase-all,
                              3 omit-first, 4 omit-last)
         byte 1       : omit count (0 unless op is 3 or 4)
         byte 2       : prefix length (0..5)
         byte 3       : suffix length (0..8)
         bytes 4..11  : prefix bytes, zero padded
         bytes 12..19 : suffix bytes, zero padded

Nothing here runs on the GPU box; the two .bin files are committed.
"""
import hashlib
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "brotli_amd", "data")

OPS = {"identity": 0, "uppercasefirst": 1, "uppercaseall": 2, "omitfirst": 3, "omitlast": 4}
RECORD = 20  # op, omit, plen, slen, prefix[8], suffix[8]


def c_unescape(s: str) -> bytes:
    out = bytearray()
    i = 0
    while i < len(s):
        ch = s[i]
        if ch != "\\":
            out.append(ord(ch))
            i += 1
            continue
        i += 1
        ch = s[i]
        if ch == "x":
            j = i + 1
            while j < len(s) and s[j] in "0123456789abcdefABCDEF":
                j += 1
            out.append(int(s[i + 1:j], 16) & 0xFF)
            i = j
            continue
        table = {"n": 10, "t": 9, '"': 34, "\\": 92, "'": 39, "r": 13}
        out.append(table[ch])
        i += 1
    return bytes(out)


def main() -> int:
    os.makedirs(OUT, exist_ok=True)
    d = open(os.path.join(REF, "brotli-dict"), "rb").read()
    want = open(os.path.join(REF, "brotli-dict.sha256")).read().split()[-1]
    assert len(d) == 122784 and hashlib.sha256(d).hexdigest() == want, "dictionary mismatch"
    open(os.path.join(OUT, "brotli_dictionary.bin"), "wb").write(d)

    txt = open(os.path.join(REF, "xforms.h")).read()
    rx = re.compile(r'/\*\s*(\d+)\s*\*/\s*\{\s*"((?:[^"\\]|\\.)*)",\s*(\w+),\s*(\d+),\s*"((?:[^"\\]|\\.)*)"\}')
    recs = bytearray()
    n = 0
    for m in rx.finditer(txt):
        idx, pre, op, omit, suf = m.groups()
        assert int(idx) == n
        p, s = c_unescape(pre), c_unescape(suf)
        assert len(p) <= 8 and len(s) <= 8
        recs += bytes([OPS[op], int(omit), len(p), len(s)]) + p.ljust(8, b"\0") + s.ljust(8, b"\0")
        n += 1
    assert n == 121, n
    open(os.path.join(OUT, "transforms.bin"), "wb").write(bytes(recs))
    print("dictionary %d bytes, transforms %d x %d bytes" % (len(d), n, RECORD))
    return 0


if __name__ == "__main__":
    sys.exit(main())

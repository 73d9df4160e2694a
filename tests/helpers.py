"""Shared test helpers: the oracle (test-only checker) and the golden fixtures."""
import ctypes
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]
_oracle = None
_ref = None


def oracle():
    """oracle/liboracle.so -- CPU restatement, test infrastructure only."""
    global _oracle
    if _oracle is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        L = ctypes.CDLL(path)
        L.oracle_yeast.restype = ctypes.c_int
        L.oracle_crc32c.restype = ctypes.c_uint32
        L.oracle_crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.oracle_xxh32.restype = ctypes.c_uint32
        L.oracle_xxh32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
        L.oracle_xxh64.restype = ctypes.c_uint64
        L.oracle_xxh64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
        L.oracle_dict_word.restype = ctypes.c_int
        L.oracle_dict_word.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
        L.oracle_context_lut.restype = ctypes.c_void_p
        L.oracle_broad.restype = ctypes.c_int
        L.oracle_broad.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        L.oracle_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        _oracle = L
    return _oracle


def reference():
    """oracle/_ref/libyeast_ref.so -- the real reference, present only where it was built."""
    global _ref
    if _ref is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libyeast_ref.so")
        if not os.path.exists(path):
            return None
        L = ctypes.CDLL(path)
        L.yeast.restype = ctypes.c_int
        L.crc32c.restype = ctypes.c_uint32
        L.crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.XXH32.restype = ctypes.c_uint32
        L.XXH32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
        L.XXH64.restype = ctypes.c_uint64
        L.XXH64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
        _ref = L
    return _ref


def run_yeast(fn, src: bytes, expected=None):
    """Call a yeast()-shaped C function.  Decode mode -> (rc, got, used, out);
    compare mode -> (rc, got, used, None)."""
    buf = ctypes.create_string_buffer(src, max(len(src), 1))
    ln = ctypes.c_size_t(len(src))
    if expected is None:
        dest = ctypes.c_void_p(None)
        got = ctypes.c_size_t(0)
        rc = fn(ctypes.byref(dest), ctypes.byref(got), buf, ctypes.byref(ln), 0)
        out = ctypes.string_at(dest.value, got.value) if dest.value else b""
        if dest.value:
            _libc.free(dest)
        return rc, got.value, ln.value, out
    keep = ctypes.create_string_buffer(expected, max(len(expected), 1))
    dest = ctypes.c_void_p(ctypes.addressof(keep))
    got = ctypes.c_size_t(len(expected))
    rc = fn(ctypes.byref(dest), ctypes.byref(got), buf, ctypes.byref(ln), 1)
    return rc, got.value, ln.value, None


def oracle_yeast(src: bytes, expected=None):
    return run_yeast(oracle().oracle_yeast, src, expected)


def load_json(name):
    return json.load(open(os.path.join(GOLD, name)))


def testdata_pairs():
    d = os.path.join(GOLD, "testdata")
    for f in sorted(os.listdir(d)):
        if f.endswith(".compressed"):
            yield f, open(os.path.join(d, f), "rb").read(), open(os.path.join(d, f[:-11]), "rb").read()


def packed_streams(name):
    meta = load_json(name + ".json")
    blob = open(os.path.join(GOLD, name + ".bin"), "rb").read()
    return meta, [blob[m["off"]:m["off"] + m["len"]] for m in meta["streams"]]


def oracle_broad(br: bytes):
    """oracle/br_oracle.c: (code, data written, streams decoded) -- the shape of brotli_amd.broad()."""
    out, n, seg = ctypes.c_void_p(None), ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = oracle().oracle_broad(br, len(br), ctypes.byref(out), ctypes.byref(n), ctypes.byref(seg))
    data = ctypes.string_at(out, n.value) if out.value and n.value else b""
    _libc.free(out)
    return rc, data, seg.value


def br_cases():
    """tests/golden/br_streams.*: (case, .br bytes); damaged copies are stored as recipes on a base case."""
    meta = load_json("br_streams.json")
    blob = open(os.path.join(GOLD, "br_streams.bin"), "rb").read()
    byname = {c["name"]: c for c in meta["cases"]}

    def data_of(c):
        if "base" in c:
            b = bytearray(data_of(byname[c["base"]]))
            if c["op"] == "xor":
                b[c["pos"]] ^= c["mask"]
                return bytes(b)
            return bytes(b[:c["pos"]])
        return blob[c["off"]:c["off"] + c["len"]]
    return [(c, data_of(c)) for c in meta["cases"]]


# Streams of this repo's own, written in the reference's brogen notation (tools/brogen.py): complex
# prefix codes, which none of the reference's .gen files describes with the `c` command.  Each is
# (text, expected output or None for "only compare decoders").
_LEAD = """w last m 4
types 1 types 1 types 1
b 2 0 b 4 0
b 2 0
types 1 types 1
"""
_TAIL = """s 1 1 10 160    # insert 4, copy 2
s 2 1 6 0
p 1 160
"""
OWN_GEN = {
    "complex-runs": (_LEAD + "c 0 3 10 3 11 3 12 3 13 4 65 4 66 4 67 4 68 5 200 5 201 5 202 5 203 5 204 5 205 5 206 5 207\n"
                     + _TAIL + "p 0 65 10 207 13\n", bytes([65, 10, 207, 13])),
    "complex-every-length": (_LEAD + "c 0 " + " ".join("%d %d" % (min(k + 1, 15), 16 * k) for k in range(16)) + "\n"
                             + _TAIL + "p 0 0 240 224 16\n", bytes([0, 240, 224, 16])),
    "complex-two-symbols": (_LEAD + "c 0 1 97 1 98\n" + _TAIL + "p 0 98 97 97 98\n", b"baab"),
    "complex-long-repeat": (_LEAD + "c 0 " + " ".join("7 %d" % k for k in range(100, 228)) + "\n"
                            + _TAIL + "p 0 100 227 163 101\n", bytes([100, 227, 163, 101])),
    "complex-cut-short": (_LEAD + "c 0 3 10 3 11 3 12 3 13 4 65 4 66 4 67 4 68 5 200 5 201 5 202 5 203 5 204 5 205 5 206 5 207\n"
                          + _TAIL + "p 0 65 10\n", None),
}


def own_streams():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from brogen import brogen
    return {k: (brogen(text), want) for k, (text, want) in OWN_GEN.items()}

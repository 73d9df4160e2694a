#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (build container only).

What it uses:
  * /root/reference/testdata/*            -- the reference's own round-trip corpus (data files,
                                             copied as fixtures; reference test: Makefile:7-8).
  * oracle/_ref/libyeast_ref.so           -- the REAL reference decoder/check functions built by
                                             oracle/Makefile from /root/reference; it produces every
                                             expected (rc, got, used, output/check) tuple stored here.
  * /usr/lib/x86_64-linux-gnu/libbrotlienc.so.1 (brotli 1.0.9, via ctypes) -- third-party ENCODER
    used only to manufacture valid streams (the reference has no compressor).  It is not part of
    parity: every stream it makes is decoded by the real reference before being stored.

Outputs (all small enough to commit):
  tests/golden/testdata/                  17 pairs copied from the reference corpus
  tests/golden/crafted.json               good/ + bad/ byte strings with the
                                          reference's rc/got/used/output
  tests/golden/cfg2_streams.bin/.json     N distinct 64 KiB q11 lgwin22 text streams (config 2 unit)
  tests/golden/misc_streams.bin/.json     whole files at other quality/lgwin/npostfix/ndirect
  tests/golden/checks.json                CRC-32C / XXH32 / XXH64 known answers
"""
import ctypes
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REFDIR = "/root/reference"

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libyeast_ref.so"))
enc = ctypes.CDLL("libbrotlienc.so.1")
libc = ctypes.CDLL(None)
libc.free.argtypes = [ctypes.c_void_p]
ref.yeast.restype = ctypes.c_int
ref.crc32c.restype = ctypes.c_uint32
ref.crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
ref.XXH32.restype = ctypes.c_uint32
ref.XXH32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
ref.XXH64.restype = ctypes.c_uint64
ref.XXH64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]

enc.BrotliEncoderCreateInstance.restype = ctypes.c_void_p
enc.BrotliEncoderCreateInstance.argtypes = [ctypes.c_void_p] * 3
enc.BrotliEncoderSetParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
enc.BrotliEncoderDestroyInstance.argtypes = [ctypes.c_void_p]
enc.BrotliEncoderCompressStream.restype = ctypes.c_int
enc.BrotliEncoderCompressStream.argtypes = [
    ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
    ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t),
    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
P_MODE, P_QUALITY, P_LGWIN, P_LGBLOCK, P_NOCTX, P_SIZE_HINT, P_LARGE, P_NPOSTFIX, P_NDIRECT = range(9)


def ref_decode(src: bytes):
    dest = ctypes.c_void_p(None)
    got = ctypes.c_size_t(0)
    ln = ctypes.c_size_t(len(src))
    buf = ctypes.create_string_buffer(src, max(len(src), 1))
    rc = ref.yeast(ctypes.byref(dest), ctypes.byref(got), buf, ctypes.byref(ln), 0)
    out = ctypes.string_at(dest.value, got.value) if dest.value else b""
    if dest.value:
        libc.free(dest)
    return rc, got.value, ln.value, out


def compress(data: bytes, quality=11, lgwin=22, npostfix=None, ndirect=None, mode=0, size_hint=True):
    st = enc.BrotliEncoderCreateInstance(None, None, None)
    enc.BrotliEncoderSetParameter(st, P_QUALITY, quality)
    enc.BrotliEncoderSetParameter(st, P_LGWIN, lgwin)
    enc.BrotliEncoderSetParameter(st, P_MODE, mode)
    if size_hint:
        enc.BrotliEncoderSetParameter(st, P_SIZE_HINT, len(data))
    if npostfix is not None:
        enc.BrotliEncoderSetParameter(st, P_NPOSTFIX, npostfix)
    if ndirect is not None:
        enc.BrotliEncoderSetParameter(st, P_NDIRECT, ndirect)
    cap = len(data) + (len(data) >> 2) + 1024
    out = ctypes.create_string_buffer(cap)
    avail_in = ctypes.c_size_t(len(data))
    next_in = ctypes.c_char_p(data)
    avail_out = ctypes.c_size_t(cap)
    next_out = ctypes.c_void_p(ctypes.addressof(out))
    total = ctypes.c_size_t(0)
    ok = enc.BrotliEncoderCompressStream(st, 2, ctypes.byref(avail_in), ctypes.byref(next_in),
                                         ctypes.byref(avail_out), ctypes.byref(next_out),
                                         ctypes.byref(total))
    enc.BrotliEncoderDestroyInstance(st)
    assert ok and avail_in.value == 0
    return out.raw[:cap - avail_out.value]


def checks(b: bytes):
    return {"crc32c": "%08x" % ref.crc32c(0, b, len(b)),
            "xxh32": "%08x" % ref.XXH32(b, len(b), 0),
            "xxh64": "%016x" % ref.XXH64(b, len(b), 0)}


def markov_text(train: bytes, n: int, seed: int) -> bytes:
    """Order-3 byte Markov chain, xorshift32."""
    table = {}
    for i in range(len(train) - 3):
        table.setdefault(train[i:i + 3], bytearray()).append(train[i + 3])
    keys = list(table.keys())
    x = seed & 0xFFFFFFFF

    def rnd():
        nonlocal x
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        return x
    out = bytearray(keys[rnd() % len(keys)])
    while len(out) < n:
        nxt = table.get(bytes(out[-3:]))
        if not nxt:
            out += keys[rnd() % len(keys)]
            continue
        out.append(nxt[rnd() % len(nxt)])
    return bytes(out[:n])


CRAFTED = {  # bytes produced by brogen from bad/*.gen, good/*.gen
    "bad/bitseof": "a20000", "bad/prefixeof": "a20000086498985b641211", "bad/metadataeof": "2c00",
    "bad/unceof": "000010", "bad/wbits": "11", "bad/end": "fe", "bad/reserved": "1c",
    "bad/metadatafill": "8c", "bad/uncfill": "0000f0", "bad/xsmetalen": "4c0000",
    "bad/xsmlen": "040000", "bad/highsymbol": "a200608401", "bad/codelenover": "a2000008b03b",
    "bad/codelenunder": "a20000080000000000", "bad/oversubscribed": "a200607017",
    "bad/incomplete": "a20060780700", "bad/toomany16": "a200600c1c67",
    "bad/toomany17": "a200600c1cf7", "bad/runlength": "a20000082224d902",
    "bad/mlenoverinsert": "020000000440401223", "bad/mlenovercopy": "220000000440201210",
    "bad/mlenoverdict": "a1280000000150801013638d0c", "bad/staticlong": "a1280000000150801013dfd303",
    "bad/staticshort": "a12800000001508010131f7a", "bad/notransform": "a128000000015080101363910c",
    "good/lastempty": "06", "good/lastzerometa": "1a",
    "good/anysimpleorder": "62000000f4581899980025006c",
    "good/anyunusedcopy": "0200000044583c16c0ffff3f", "good/xsblockcount": "020020a20003001116880400",
    "good/anysinglelen": "1000000070000000000424018800000080010000001090042002000000040000002020"
                         "094004000000040000004040128011000000780000000002920004",
}


def main():
    os.makedirs(os.path.join(GOLD, "testdata"), exist_ok=True)
    # 1. the reference corpus
    for f in sorted(glob.glob(os.path.join(REFDIR, "testdata", "*"))):
        shutil.copyfile(f, os.path.join(GOLD, "testdata", os.path.basename(f)))
    idx = {}
    for c in sorted(glob.glob(os.path.join(GOLD, "testdata", "*.compressed"))):
        src = open(c, "rb").read()
        orig = open(c[:-len(".compressed")], "rb").read()
        rc, got, used, out = ref_decode(src)
        assert rc == 0 and out == orig, c
        idx[os.path.basename(c)] = dict(rc=rc, got=got, used=used, **checks(out))
    json.dump(idx, open(os.path.join(GOLD, "testdata.json"), "w"), indent=1, sort_keys=True)

    # 2. crafted streams
    crafted = {}
    for name, hx in CRAFTED.items():
        rc, got, used, out = ref_decode(bytes.fromhex(hx))
        crafted[name] = dict(hex=hx, rc=rc, got=got, used=used, out=out.hex())
    json.dump(crafted, open(os.path.join(GOLD, "crafted.json"), "w"), indent=1, sort_keys=True)

    # 3. config-2 unit streams: 64 KiB windows of English text, q11, lgwin 22
    names = ["alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"]
    text = b"".join(open(os.path.join(REFDIR, "testdata", n), "rb").read() for n in names)
    CH = 65536
    n_real, n_markov = 192, 64
    stride = (len(text) - CH) // (n_real - 1)
    chunks = [("text", i * stride, text[i * stride:i * stride + CH]) for i in range(n_real)]
    mk = markov_text(text, n_markov * CH, 0x5EED)
    chunks += [("markov", i * CH, mk[i * CH:(i + 1) * CH]) for i in range(n_markov)]
    blob = bytearray()
    meta = []
    for kind, off, data in chunks:
        c = compress(data, 11, 22)
        rc, got, used, out = ref_decode(c)
        assert rc == 0 and out == data and used == len(c)
        meta.append(dict(kind=kind, src_off=off, off=len(blob), len=len(c), got=got, **checks(out)))
        blob += c
    open(os.path.join(GOLD, "cfg2_streams.bin"), "wb").write(bytes(blob))
    json.dump(dict(corpus=names, chunk=CH, quality=11, lgwin=22, encoder="libbrotlienc 1.0.9",
                   markov_seed=0x5EED, streams=meta),
              open(os.path.join(GOLD, "cfg2_streams.json"), "w"), indent=0)
    print("cfg2: %d streams, %d bytes compressed" % (len(meta), len(blob)))

    # 4. misc: other parameters on whole files (smaller ones to keep the repo light)
    blob = bytearray()
    meta = []
    alice = open(os.path.join(REFDIR, "testdata", "alice29.txt"), "rb").read()
    asyou = open(os.path.join(REFDIR, "testdata", "asyoulik.txt"), "rb").read()
    variants = []
    for q in (0, 1, 2, 4, 5, 6, 9, 10):
        variants.append(("alice29.txt", alice, dict(quality=q, lgwin=22)))
    for lg in (10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24):
        variants.append(("asyoulik.txt", asyou, dict(quality=9, lgwin=lg)))
    for npf, ndr in ((1, 0), (2, 0), (3, 0), (0, 15), (1, 30), (2, 60), (3, 120), (3, 8), (1, 2)):
        variants.append(("alice29.txt", alice[:60000], dict(quality=11, lgwin=18, npostfix=npf, ndirect=ndr)))
    for mode in (1, 2):  # text / font modes change the context modes the encoder picks
        variants.append(("asyoulik.txt", asyou[:50000], dict(quality=11, lgwin=16, mode=mode)))
    for name, data, kw in variants:
        c = compress(data, **kw)
        rc, got, used, out = ref_decode(c)
        assert rc == 0 and out == data, (name, kw)
        meta.append(dict(file=name, n=len(data), params=kw, off=len(blob), len=len(c), got=got,
                         used=used, **checks(out)))
        blob += c
    open(os.path.join(GOLD, "misc_streams.bin"), "wb").write(bytes(blob))
    json.dump(dict(encoder="libbrotlienc 1.0.9", streams=meta),
              open(os.path.join(GOLD, "misc_streams.json"), "w"), indent=0)
    print("misc: %d streams, %d bytes" % (len(meta), len(blob)))

    # 5. check-value known answers (SURVEY section 4 item 3 + a few lengths around the lane widths)
    kat = {"123456789": checks(b"123456789"), "empty": checks(b"")}
    kat["alice29.txt"] = checks(alice)
    assert kat["alice29.txt"] == {"crc32c": "ebd73954", "xxh32": "d0313f4a", "xxh64": "f4530439148be73a"}
    for n in (1, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 1000, 4095, 4096, 4097, 65536, 100000):
        kat["alice29.txt[:%d]" % n] = checks(alice[:n])
    json.dump(kat, open(os.path.join(GOLD, "checks.json"), "w"), indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""One-off soak with REAL encoder output: libbrotlienc 1.0.9 (the image's own, a fixture generator only -- DESIGN.md section 2)
compresses varied data at random quality, window, block size and flush points (so: many meta-blocks, block switching, context
maps, dictionary words, blocks that end in copies, empty padding blocks), and every kernel form decodes the streams and their
damaged copies; rc, got, used and the output bytes against the CPU restatement (which tests/test_oracle_golden.py pins to the
reference).
usage: tools/soak_encoder.py [seeds] [streams per seed] [first seed]"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H
import brotli_amd as ba

enc = ctypes.CDLL("libbrotlienc.so.1")
enc.BrotliEncoderCreateInstance.restype = ctypes.c_void_p
enc.BrotliEncoderCreateInstance.argtypes = [ctypes.c_void_p] * 3
enc.BrotliEncoderSetParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
enc.BrotliEncoderDestroyInstance.argtypes = [ctypes.c_void_p]
enc.BrotliEncoderCompressStream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                            ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t),
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
enc.BrotliEncoderHasMoreOutput.argtypes = [ctypes.c_void_p]
enc.BrotliEncoderTakeOutput.restype = ctypes.c_void_p
enc.BrotliEncoderTakeOutput.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
PROCESS, FLUSH, FINISH = 0, 1, 2
P_MODE, P_QUALITY, P_LGWIN, P_LGBLOCK = 0, 1, 2, 3


def compress(data, rng):
    st = enc.BrotliEncoderCreateInstance(None, None, None)
    q = rng.choice([0, 1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 9, 10, 10, 11, 11])
    enc.BrotliEncoderSetParameter(st, P_QUALITY, q)
    enc.BrotliEncoderSetParameter(st, P_LGWIN, rng.choice([10, 12, 16, 18, 20, 22, 24]))
    enc.BrotliEncoderSetParameter(st, P_MODE, rng.choice([0, 0, 1, 2]))
    if rng.random() < 0.5:
        enc.BrotliEncoderSetParameter(st, P_LGBLOCK, rng.choice([16, 17, 18]))
    out = bytearray()

    def drain():
        while True:
            n = ctypes.c_size_t(0)
            p = enc.BrotliEncoderTakeOutput(st, ctypes.byref(n))
            if not n.value:
                break
            out.extend(ctypes.string_at(p, n.value))
    pos = 0
    while True:
        step = rng.choice([1, 7, 300, 5000, 70000, len(data)])
        piece = data[pos:pos + step]
        pos += len(piece)
        last = pos >= len(data)
        op = FINISH if last else (FLUSH if rng.random() < 0.3 else PROCESS)
        avail_in = ctypes.c_size_t(len(piece))
        nxt = ctypes.c_char_p(piece)
        avail_out = ctypes.c_size_t(0)
        while True:
            ok = enc.BrotliEncoderCompressStream(st, op, ctypes.byref(avail_in), ctypes.byref(nxt), ctypes.byref(avail_out), None, None)
            assert ok
            drain()
            if avail_in.value == 0 and not enc.BrotliEncoderHasMoreOutput(st):
                break
        if last:
            break
    drain()
    enc.BrotliEncoderDestroyInstance(st)
    return bytes(out)


def sample(rng, corpus):
    kind = rng.random()
    n = rng.choice([0, 1, 5, 60, 700, 4000, 20000, 70000, 200000])
    if kind < 0.45:
        src = rng.choice(corpus)
        a = rng.randrange(0, max(1, len(src) - n))
        return src[a:a + n]
    if kind < 0.6:
        return bytes(rng.getrandbits(8) for _ in range(min(n, 20000)))
    if kind < 0.8:  # repetitive: short periods, long runs
        unit = bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 2, 3, 7, 16, 31, 100])))
        return (unit * (n // len(unit) + 1))[:n]
    a, b = rng.choice(corpus), rng.choice(corpus)  # text with binary islands
    out = bytearray()
    while len(out) < n:
        s = rng.choice([a, b])
        p = rng.randrange(0, max(1, len(s) - 500))
        out += s[p:p + rng.randrange(1, 500)]
        if rng.random() < 0.3:
            out += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 60)))
    return bytes(out[:n])


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 7000
    corpus = [orig for _, _, orig in H.testdata_pairs() if len(orig) > 1000]
    bad = 0
    for seed in range(first, first + seeds):
        t0 = time.time()
        rng = random.Random(seed)
        ss = [compress(sample(rng, corpus), rng) for _ in range(per)]
        extra = []
        for s in ss:
            b = bytearray(s)
            if rng.random() < 0.3:
                b = b[:rng.randrange(1, len(b) + 1)]
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            extra.append(bytes(b))
        ss += extra
        want = [H.oracle_yeast(s) for s in ss]
        t1 = time.time()
        for lanes, merge, half in ((1, 3, 1), (1, 3, 0), (1, 4, 1), (1, 4, 0), (1, 0, 1), (1, 1, 1), (1, 2, 1), (1, 2, 0), (65, 3, 1), (64, 3, 1), (16, 3, 1)):
            os.environ["YEAST_FORCE_LANES"] = str(lanes)
            os.environ["YEAST_MERGE"] = str(merge)
            os.environ["YEAST_HALF"] = str(half)
            res = ba.yeast_batch(ss, ba.CHECK_NONE)
            for i, (w, r) in enumerate(zip(want, res)):
                if (r[0], r[2], r[3], r[1]) != (w[0], w[1], w[2], w[3]):
                    bad += 1
                    print("MISMATCH seed %d stream %d lanes %d merge %d half %d: gpu %s oracle %s" % (seed, i, lanes, merge, half, (r[0], r[2], r[3]), w[:3]), flush=True)
        # compare mode (yeast(cmp = 1): the cooperative kernel's other instantiation) on the intact streams: the exact data, a
        # flipped byte, data cut short and data with a tail; (rc, got, used) as the restatement reports them
        for k in ("YEAST_FORCE_LANES", "YEAST_MERGE", "YEAST_HALF"):
            os.environ.pop(k, None)
        for i in range(0, per, 3):
            s, data = ss[i], want[i][3]
            if want[i][0] != 0:
                continue
            exps = [data, data + b"x"]
            if len(data) > 1:
                e = bytearray(data)
                e[rng.randrange(len(e))] ^= 0x10
                exps += [bytes(e), data[:rng.randrange(0, len(data))]]
            for e in exps:
                rc, out, got, used = ba.yeast(s, e)
                w = H.oracle_yeast(s, e)
                if (rc, got, used) != w[:3]:
                    bad += 1
                    print("MISMATCH (compare mode) seed %d stream %d: gpu %s oracle %s" % (seed, i, (rc, got, used), w[:3]), flush=True)
        print("seed %d: %d streams (%d bytes out), encode+oracle %.1f s, gpu %.1f s" % (seed, len(ss), sum(w[1] for w in want), t1 - t0, time.time() - t1), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

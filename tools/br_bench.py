#!/usr/bin/env python3
"""Throughput of the .br reader (yeast_broad) on a seekable file of N segments, each one
64 KiB text stream of the cfg2 fixtures with an XXH64 check, reverse offsets, total length and an XXH32
check of checks -- against this repo's CPU restatement of broad() on one host core (and the reference's
own broad() where oracle/_ref exists).  Host buffers in, host buffer out: PCIe and the host-side walk
are inside the timed region.  usage: tools/br_bench.py [segments]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import brotli_amd as ba


def par(n):
    return (0x34cb00 >> ((n ^ (n >> 4)) & 0xf)) & 0x80


def var(n):
    out = bytearray()
    while n > 0x7f:
        out.append(n & 0x7f)
        n >>= 7
    out.append(0x80 | n)
    return bytes(out)


def bvar(n):
    out = bytearray([0x80 | (n & 0x7f)])
    n >>= 7
    while n > 0x7f:
        out.append(n & 0x7f)
        n >>= 7
    out.append(0x80 | n)
    return bytes(out)


def build(nseg):
    meta, streams = H.packed_streams("cfg2_streams")
    out = bytearray(b"\xce\xb2\xcf\x81")
    last, checks, total = 0, bytearray(), 0
    for k in range(nseg):
        j = (k * 37) % len(streams)
        here = len(out)
        mask = 3 | (0x10 if last else 0)
        out.append(mask ^ par(mask))
        if last:
            out += var(here - last)
        out += streams[j]
        c = int(meta["streams"][j]["xxh64"], 16).to_bytes(8, "little")
        out += c
        checks += c
        total += meta["streams"][j]["got"]
        last = here
    mask = 0x20 | 2 | 8 | 0x10
    mask ^= par(mask)
    out.append(mask)
    out += bvar(len(out) - 1 - last) + bvar(total) + ba.xxh32(bytes(checks)).to_bytes(4, "little")
    out.append(mask)
    return bytes(out), total


def main():
    nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    data, total = build(nseg)
    assert ba.broad(data, write=False) == (0, b"", nseg)
    res = {"segments": nseg, "br_bytes": len(data), "out_bytes": total}
    import ctypes
    L = ba.lib()
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    for write in (False, True):  # the C call alone (no copy into a Python object), best of three
        best = None
        for _ in range(3):
            out, n, seg = ctypes.c_void_p(None), ctypes.c_size_t(0), ctypes.c_size_t(0)
            t0 = time.perf_counter()
            rc = L.yeast_broad(data, len(data), ctypes.byref(out), ctypes.byref(n), 1 if write else 0, ctypes.byref(seg))
            dt = time.perf_counter() - t0
            assert rc == 0 and seg.value == nseg and n.value == (total if write else 0)
            libc.free(out)
            best = dt if best is None or dt < best else best
        res["gpu_verify_only_GBps" if not write else "gpu_unwrap_GBps"] = total / best / 1e9
    small, stotal = build(min(nseg, 256))
    t0 = time.perf_counter()
    rc, out, seg = H.oracle_broad(small)
    dt = time.perf_counter() - t0
    assert rc == 0 and len(out) == stotal
    res["cpu_restatement_1core_GBps"] = stotal / dt / 1e9
    ref = os.path.join(ROOT, "oracle", "_ref", "libbroad_ref.so")
    res["cpu_reference_1core_GBps"] = None
    if os.path.exists(ref):
        import ctypes, tempfile
        L = ctypes.CDLL(ref)
        libc = ctypes.CDLL(None)
        libc.fopen.restype = ctypes.c_void_p
        libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        libc.fclose.argtypes = [ctypes.c_void_p]
        L.broad.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        d = tempfile.mkdtemp()
        open(os.path.join(d, "i"), "wb").write(small)
        i, o = libc.fopen(os.path.join(d, "i").encode(), b"rb"), libc.fopen(os.path.join(d, "o").encode(), b"wb")
        t0 = time.perf_counter()
        rc = L.broad(i, o, 0, 1)
        dt = time.perf_counter() - t0
        libc.fclose(i)
        libc.fclose(o)
        assert rc == 0
        res["cpu_reference_1core_GBps"] = stotal / dt / 1e9
    print(json.dumps(res))


if __name__ == "__main__":
    main()

"""brotli_amd -- host-side mirror of the reference's decode interface on top of
libyeast_mi355x.so (HIP kernels for MI355X / gfx950).

The functions keep the names, argument meaning and error behaviour of the
reference interface they stand for (paths relative to /root/reference):

  yeast(source, expected=None)    yeast.h:25 / yeast.c:1353  -> (rc, out, used)
  yeast_batch(streams, check)     a loop over yeast.h:25 such as juxt.c:77-93
  crc32c / xxh32 / xxh64          crc32c.c:258, xxhash.c:365, xxhash.c:505
  broad(br, write=True)           broad.c:142 over a .br file in memory -> (code, data, streams)
  brand(stream, opt, name)        brand.c:286 (wrap one stream)          -> (status, .br bytes)
  braid(files)                    braid.c:341 (merge .br files)          -> (exit status, .br bytes)

There is no CPU decode path: importing works anywhere (so the CPU test tier
can check that the library loads and exports its symbols), but every compute
call raises RuntimeError when the HIP library or a GPU is missing.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# YEAST_LIB selects another build of the same library (tools/phase_prof.py uses an instrumented one)
LIB_PATH = os.environ.get("YEAST_LIB") or os.path.join(_HERE, "libyeast_mi355x.so")

CHECK_NONE, CHECK_CRC32C, CHECK_XXH32, CHECK_XXH64 = 0, 1, 2, 3
SRC_PAD = 32

EXPORTS = [  # every symbol include/yeast.h, include/yeast_batch.h and include/yeast_br.h declare
    "yeast", "yeast_verbosity", "yeast_batch", "yeast_batch_device", "yeast_check_device",
    "crc32c", "XXH32", "XXH64", "yeast_last_error", "yeast_device_count", "yeast_debug_table", "yeast_trim",
    "XXH32_createState", "XXH32_freeState", "XXH32_reset", "XXH32_update", "XXH32_digest",
    "XXH64_createState", "XXH64_freeState", "XXH64_reset", "XXH64_update", "XXH64_digest",
    "yeast_broad", "yeast_brand", "yeast_braid", "yeast_sha256", "yeast_sha256_batch",  # include/yeast_br.h
]


class Record(ctypes.Structure):
    """yeast_record of include/yeast_batch.h (32 bytes)."""
    _fields_ = [("rc", ctypes.c_int32), ("flags", ctypes.c_uint32), ("got", ctypes.c_uint64),
                ("used", ctypes.c_uint64), ("check", ctypes.c_uint64)]


_lib = None
_knobs = None
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def lib() -> ctypes.CDLL:
    """Load libyeast_mi355x.so; raise loudly if it has not been built."""
    global _lib, _knobs
    if _lib is not None:
        # the library reads its YEAST_* knobs once; tests and tools that change them between calls say so
        now = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("YEAST_")))
        if now != _knobs:
            _knobs = now
            _lib.yeast_reload_knobs()
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)" % LIB_PATH)
        # torch (used by brotli_amd.device for HBM buffers and streams) ships a HIP runtime of its own under the
        # same soname: whichever is loaded first serves the whole process, and torch finds no GPU behind the
        # system one.  So torch goes first whenever it is there; the C library itself never needs it.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.yeast_reload_knobs.restype = None
        _knobs = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("YEAST_")))
        L.yeast.restype = ctypes.c_int
        L.yeast.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                            ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]
        L.yeast_batch.restype = ctypes.c_int
        L.yeast_batch.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.POINTER(ctypes.c_int), ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64)]
        L.yeast_batch_device.restype = ctypes.c_int
        L.yeast_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int,
                                         ctypes.c_void_p]
        L.yeast_batch_device_static.restype = ctypes.c_int
        L.yeast_batch_device_static.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p]
        L.yeast_init.restype = ctypes.c_int
        L.yeast_check_device.restype = ctypes.c_int
        L.yeast_check_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_uint, ctypes.c_void_p]
        L.crc32c.restype = ctypes.c_uint32
        L.crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.XXH32.restype = ctypes.c_uint
        L.XXH32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint]
        L.XXH64.restype = ctypes.c_ulonglong
        L.XXH64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_ulonglong]
        L.yeast_last_error.restype = ctypes.c_char_p
        L.yeast_last_kernel.restype = ctypes.c_char_p
        L.yeast_device_count.restype = ctypes.c_int
        L.yeast_debug_table.restype = ctypes.c_void_p
        L.yeast_debug_table.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
        L.yeast_broad.restype = ctypes.c_int
        L.yeast_broad.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p),
                                  ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
        L.yeast_brand.restype = ctypes.c_int
        L.yeast_brand.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p,
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        L.yeast_braid.restype = ctypes.c_int
        L.yeast_braid.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        for bits, seed_t, out_t in ((32, ctypes.c_uint, ctypes.c_uint), (64, ctypes.c_ulonglong, ctypes.c_ulonglong)):
            getattr(L, "XXH%d_createState" % bits).restype = ctypes.c_void_p
            getattr(L, "XXH%d_freeState" % bits).argtypes = [ctypes.c_void_p]
            getattr(L, "XXH%d_reset" % bits).argtypes = [ctypes.c_void_p, seed_t]
            getattr(L, "XXH%d_update" % bits).argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
            getattr(L, "XXH%d_digest" % bits).argtypes = [ctypes.c_void_p]
            getattr(L, "XXH%d_digest" % bits).restype = out_t
        L.yeast_sha256.restype = ctypes.c_int
        L.yeast_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        _lib = L
    return _lib


def _err(what: str) -> RuntimeError:
    return RuntimeError("%s: %s" % (what, lib().yeast_last_error().decode() or "GPU path unavailable"))


def yeast(source: bytes, expected: Optional[bytes] = None) -> Tuple[int, Optional[bytes], int, int]:
    """yeast.h:25.  Decode mode (expected is None): returns (rc, output, got, used).
    Compare mode: returns (rc, None, got, used) where got is how far the output matched."""
    L = lib()
    src = ctypes.create_string_buffer(source, max(len(source), 1))
    ln = ctypes.c_size_t(len(source))
    if expected is None:
        dest = ctypes.c_void_p(None)
        got = ctypes.c_size_t(0)
        rc = L.yeast(ctypes.byref(dest), ctypes.byref(got), src, ctypes.byref(ln), 0)
        if rc < 0:
            raise _err("yeast")
        out = ctypes.string_at(dest.value, got.value) if dest.value else b""
        if dest.value:
            _libc.free(dest)
        return rc, out, got.value, ln.value
    keep = ctypes.create_string_buffer(expected, max(len(expected), 1))
    dest = ctypes.c_void_p(ctypes.addressof(keep))
    got = ctypes.c_size_t(len(expected))
    rc = L.yeast(ctypes.byref(dest), ctypes.byref(got), src, ctypes.byref(ln), 1)
    if rc < 0:
        raise _err("yeast")
    return rc, None, got.value, ln.value


def yeast_batch(streams: Sequence[bytes], check: int = CHECK_NONE):
    """Decode independent streams in one launch.  Returns a list of
    (rc, output, got, used, check) -- the tuple yeast() would give per stream plus the check."""
    L = lib()
    n = len(streams)
    if n == 0:
        return []
    bufs = [ctypes.create_string_buffer(s, max(len(s), 1)) for s in streams]
    src = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in bufs])
    ln = (ctypes.c_size_t * n)(*[len(s) for s in streams])
    dest = (ctypes.c_void_p * n)()
    got = (ctypes.c_size_t * n)()
    rc = (ctypes.c_int * n)()
    chk = (ctypes.c_uint64 * n)()
    e = L.yeast_batch(n, src, ln, dest, got, rc, check, chk)
    if e:
        raise _err("yeast_batch")
    out = []
    for i in range(n):
        data = ctypes.string_at(dest[i], got[i]) if dest[i] else b""
        if dest[i]:
            _libc.free(dest[i])
        out.append((rc[i], data, got[i], ln[i], chk[i]))
    return out


def crc32c(data: bytes, crc: int = 0) -> int:
    return lib().crc32c(crc, data, len(data))


def xxh32(data: bytes, seed: int = 0) -> int:
    return lib().XXH32(data, len(data), seed)


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().XXH64(data, len(data), seed)


def broad(br: bytes, write: bool = True) -> Tuple[int, bytes, int]:
    """broad() of broad.c:142-353 over a .br file held in memory: (code, data written, streams decoded).
    Codes 0 ok, 2 premature end, 3 invalid format, 4 invalid compressed data, 5 length/check mismatch."""
    out, n, seg = ctypes.c_void_p(None), ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = lib().yeast_broad(br, len(br), ctypes.byref(out), ctypes.byref(n), 1 if write else 0, ctypes.byref(seg))
    if rc < 0:
        raise _err("yeast_broad")
    data = ctypes.string_at(out, n.value) if out.value and n.value else b""
    _libc.free(out)
    return rc, data, seg.value


def brand(stream: bytes, opt: str = "", name: Optional[str] = None) -> Tuple[int, bytes]:
    """The reference's brand program (brand.c:286-314): wrap one brotli stream -> (status, .br bytes)."""
    out, n = ctypes.c_void_p(None), ctypes.c_size_t(0)
    rc = lib().yeast_brand(stream, len(stream), opt.encode(), name.encode() if name is not None else None,
                           ctypes.byref(out), ctypes.byref(n))
    if rc < 0:
        raise _err("yeast_brand")
    data = ctypes.string_at(out, n.value) if out.value and n.value else b""
    _libc.free(out)
    return rc, data


def braid(files: Sequence[bytes]) -> Tuple[int, bytes]:
    """The reference's braid program (braid.c:341-422): merge .br files -> (exit status, .br bytes)."""
    k = len(files)
    arr = (ctypes.c_char_p * max(k, 1))(*files)
    lens = (ctypes.c_size_t * max(k, 1))(*[len(f) for f in files])
    out, n = ctypes.c_void_p(None), ctypes.c_size_t(0)
    rc = lib().yeast_braid(k, arr, lens, ctypes.byref(out), ctypes.byref(n))
    if rc < 0:
        raise _err("yeast_braid")
    data = ctypes.string_at(out, n.value) if out.value and n.value else b""
    _libc.free(out)
    return rc, data


class XXHState:
    """The reference's streaming hash interface (xxhash.h:149-167): reset / update / digest."""

    def __init__(self, bits: int = 64, seed: int = 0):
        assert bits in (32, 64)
        self._f = lambda name: getattr(lib(), "XXH%d_%s" % (bits, name))
        self._s = self._f("createState")()
        if not self._s:
            raise MemoryError("XXH%d_createState" % bits)
        self.reset(seed)

    def reset(self, seed: int = 0):
        self._f("reset")(self._s, seed)
        return self

    def update(self, data: bytes):
        if self._f("update")(self._s, data, len(data)):
            raise _err("XXH_update")
        return self

    def digest(self) -> int:
        return self._f("digest")(self._s)

    def __del__(self):
        if getattr(self, "_s", None):
            self._f("freeState")(self._s)
            self._s = None


def sha256(data: bytes) -> bytes:
    """SHA-256 on the GPU (the reference: OpenSSL's SHA256(), broad.c:266)."""
    out = ctypes.create_string_buffer(32)
    if lib().yeast_sha256(data, len(data), out):
        raise _err("yeast_sha256")
    return out.raw


def trim() -> None:
    """yeast_trim(): give back the buffers the library holds on the current device between calls."""
    if lib().yeast_trim():
        raise _err("yeast_trim")


def debug_table(which: int) -> bytes:
    sz = ctypes.c_size_t(0)
    p = lib().yeast_debug_table(which, ctypes.byref(sz))
    return ctypes.string_at(p, sz.value) if p else b""

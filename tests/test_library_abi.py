"""CPU tier: the C-ABI library loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re
import zlib

import brotli_amd as ba
import helpers as H


def declared_symbols():
    syms = set()
    for h in ("yeast.h", "yeast_batch.h", "yeast_br.h"):
        txt = open(os.path.join(H.ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"\b(\w+)\s*\(", txt):
            if m.group(1) not in ("defined", "if"):
                syms.add(m.group(1))
        for m in re.finditer(r"extern\s+int\s+(\w+)\s*;", txt):
            syms.add(m.group(1))
    return syms


def test_library_exports_every_declared_symbol():
    L = ba.lib()
    decl = declared_symbols()
    assert set(ba.EXPORTS) <= decl, set(ba.EXPORTS) - decl
    for s in decl:
        assert hasattr(L, s), "missing export %s" % s
    assert ctypes.sizeof(ba.Record) == 32


def test_static_tables_known_answers():
    # check.c:94-99: zlib CRC-32 of lut0/lut1/lut2, the dictionary and the transforms
    assert zlib.crc32(ba.debug_table(0)) == 0x8e91efb7
    assert zlib.crc32(ba.debug_table(1)) == 0xd01a32f4
    assert zlib.crc32(ba.debug_table(2)) == 0x0dd7a0d6
    assert len(ba.debug_table(3)) == 122784 and zlib.crc32(ba.debug_table(3)) == 0x5136cb04
    assert len(ba.debug_table(4)) == 121 * 20


_NO_DEVICE = r"""
import ctypes, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import brotli_amd as ba, helpers as H
L = ba.lib()
# 1. yeast_batch() with device set-up failing: dest[] is a pure output, so whatever the caller left in it must be
#    neither read nor freed (a poisoned pointer would abort the process in free()); the call returns -1
n = 3
s = bytes.fromhex("a20000")
bufs = [ctypes.create_string_buffer(s, len(s)) for _ in range(n)]
src = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in bufs])
ln = (ctypes.c_size_t * n)(3, 3, 3)
dest = (ctypes.c_void_p * n)(0x5a5a5a5a5a5a5a50, 0xdeadbeefdeadbee0, 0x10)
got = (ctypes.c_size_t * n)()
rc = (ctypes.c_int * n)()
chk = (ctypes.c_uint64 * n)()
e = L.yeast_batch(n, src, ln, dest, got, rc, 0, chk)
assert e == -1, e
assert [dest[i] for i in range(n)] == [None] * n, [dest[i] for i in range(n)]
# 2. a check value that cannot be computed is never written or compared as 0: the merger of two files needs
#    XXH32 of the copied checks for its trailer and must fail, and the reader must not accept a stored 0
cases = [(c, d) for c, d in H.br_cases() if "base" not in c and c.get("rc") == 0 and c["name"].startswith("single/")]
files = [d for c, d in cases[:2]]
k = len(files)
arr = (ctypes.c_char_p * k)(*files)
lens = (ctypes.c_size_t * k)(*[len(f) for f in files])
out, m = ctypes.c_void_p(None), ctypes.c_size_t(0)
e = L.yeast_braid(k, arr, lens, ctypes.byref(out), ctypes.byref(m))
assert e == -1 and not out.value, (e, out.value)
assert L.yeast_check_error() == 0  # read and cleared by the library itself
assert L.XXH32(b"abc", 3, 0) == 0 and L.yeast_check_error() == 1 and L.yeast_check_error() == 0
print("no-device ok")
"""


def test_failures_of_device_setup_leave_nothing_wrong_behind():
    """with no usable GPU, yeast_batch() must not free the caller's dest[] contents, and the .br
    writer/merger must fail instead of writing a check value of 0 (a child process with the devices hidden)."""
    import subprocess
    import sys
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", _NO_DEVICE % (H.ROOT, os.path.join(H.ROOT, "tests"))],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "no-device ok" in out.stdout, out.stdout + out.stderr

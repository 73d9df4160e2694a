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

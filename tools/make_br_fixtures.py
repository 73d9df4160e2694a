#!/usr/bin/env python3
"""Golden vectors for the .br container reader (reference: broad.c:142-353).

Runs in the build container only.  The .br files are written by the reference's own brand and braid
programs and judged by the reference's own broad(), all three compiled from the sources where they lie
under /root/reference (oracle/Makefile, target `brref`: plain gcc on those files plus the image's
libcrypto for SHA-256).  Output: tests/golden/br_streams.bin + br_streams.json -- for every case the
.br bytes, the code broad() returns, and the length and XXH64 of what it wrote (the bytes themselves when
short).  Nothing of the reference is copied: the fixtures are inputs and expected outputs.
"""
import ctypes, json, os, random, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H

REF = os.path.join(ROOT, "oracle", "_ref")
TD = os.path.join(ROOT, "tests", "golden", "testdata")
ref = ctypes.CDLL(os.path.join(REF, "libbroad_ref.so"))
libc = ctypes.CDLL(None)
libc.fopen.restype = ctypes.c_void_p
libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
libc.fclose.argtypes = [ctypes.c_void_p]
ref.broad.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
tmp = tempfile.mkdtemp()


def ref_broad(data):
    pi, po = os.path.join(tmp, "in.br"), os.path.join(tmp, "out.bin")
    open(pi, "wb").write(data)
    i, o = libc.fopen(pi.encode(), b"rb"), libc.fopen(po.encode(), b"wb")
    err = os.dup(2)
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 2)
    try:
        rc = ref.broad(i, o, 0, 1)
    finally:
        os.dup2(err, 2)
        os.close(null)
        os.close(err)
    libc.fclose(i)
    libc.fclose(o)
    return rc, open(po, "rb").read()


def brand(stream, opt="", name=None):
    args = [os.path.join(REF, "brand")] + ([opt] if opt or name else []) + ([name] if name else [])
    return subprocess.run(args, input=stream, stdout=subprocess.PIPE, check=True).stdout


def braid(files):
    paths = []
    for k, f in enumerate(files):
        p = os.path.join(tmp, "b%d.br" % k)
        open(p, "wb").write(f)
        paths.append(p)
    return subprocess.run([os.path.join(REF, "braid")] + paths, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout


def td(name):
    return open(os.path.join(TD, name + ".compressed"), "rb").read()


cases = []
made = {}     # name -> how the writer/merger tests rebuild the file: {"brand": [testdata stream, options, name]},
              # {"braid": [case names]} or {"braid_brand": [[kind, stream, options], ...]}; "ret" = braid's exit status
recipes = {}  # name -> (base name, "xor", pos, mask) or (base name, "cut", n): damaged copies are stored as recipes


def add(name, data, recipe=None):
    rc, out = ref_broad(data)
    cases.append((name, data, rc, out))
    if recipe:
        recipes[name] = recipe


small = ["quickfox", "x", "xyzzy", "10x10y", "64x", "empty", "monkey", "ukkonooa"]
opts = ["", "x4", "x2", "x1", "c", "c4", "c2", "c1", "s", "n", "u", "r", "b", "xn", "c2u", "sr", "f", "x4f"]
singles = {}
for o in opts:
    for s in (small if o in ("", "c", "s") else small[:3]):
        b = brand(td(s), o, "a name" if "f" in o else None)
        singles[(s, o)] = b
        add("single/%s/%s" % (s, o or "default"), b)
        made["single/%s/%s" % (s, o or "default")] = {"brand": [s, o, "a name" if "f" in o else None]}
# the two streams of br-format-v3.txt:416-429
add("spec/shortest-with-data", bytes.fromhex("ceb2cf8184060027"))
add("spec/shortest", bytes.fromhex("ceb2cf8127"))
# large outputs
for s in ("alice29.txt", "zeros", "backward65536", "compressed_repeated"):
    for o_ in ("", "c", "s"):
        add("single/%s/%s" % (s, o_ or "default"), brand(td(s), o_))
        made["single/%s/%s" % (s, o_ or "default")] = {"brand": [s, o_, None]}

# merged files (braid.c): reverse offsets everywhere, total length, XXH32 check of checks
multis = {}
multis["three-kinds"] = braid([singles[("quickfox", "c2")], singles[("x", "s")], singles[("xyzzy", "x4f")]])
multis["eight-small"] = braid([singles[(s, "")] for s in small])
multis["eight-crc"] = braid([singles[(s, "c")] for s in small])
multis["eight-sha"] = braid([singles[(s, "s")] for s in small])
multis["mixed-16"] = braid([singles[(s, o)] for o in ("", "c", "s") for s in small[:5]] + [singles[("x", "x1")]])
multis["merged-of-merged"] = braid([multis["three-kinds"], multis["eight-small"], singles[("monkey", "c")]])
meta, cfg2 = H.packed_streams("cfg2_streams")
multis["text-12x64KiB"] = braid([brand(cfg2[k], ("", "c", "x4", "s")[k % 4]) for k in range(12)])
multis["large-mix"] = braid([brand(td("alice29.txt")), brand(td("zeros"), "c"), brand(td("backward65536"), "s"),
                             brand(td("quickfox_repeated"), "x2")])
multis["no-length"] = braid([singles[("quickfox", "n")], singles[("x", "r")], singles[("xyzzy", "")]])
def nm(s_, o_):
    return "single/%s/%s" % (s_, o_ or "default")
made["multi/three-kinds"] = {"braid": [nm("quickfox", "c2"), nm("x", "s"), nm("xyzzy", "x4f")]}
made["multi/eight-small"] = {"braid": [nm(s, "") for s in small]}
made["multi/eight-crc"] = {"braid": [nm(s, "c") for s in small]}
made["multi/eight-sha"] = {"braid": [nm(s, "s") for s in small]}
made["multi/mixed-16"] = {"braid": [nm(s, o_) for o_ in ("", "c", "s") for s in small[:5]] + [nm("x", "x1")]}
made["multi/merged-of-merged"] = {"braid": ["multi/three-kinds", "multi/eight-small", nm("monkey", "c")]}
made["multi/text-12x64KiB"] = {"braid_brand": [["cfg2", k, ("", "c", "x4", "s")[k % 4]] for k in range(12)]}
made["multi/large-mix"] = {"braid_brand": [["td", "alice29.txt", ""], ["td", "zeros", "c"], ["td", "backward65536", "s"],
                                            ["td", "quickfox_repeated", "x2"]]}
made["multi/no-length"] = {"braid": [nm("quickfox", "n"), nm("x", "r"), nm("xyzzy", "")]}
# files braid skips (no chain of reverse offsets, damage) among good ones, and the empty merge
def braid_rc(files):
    paths = []
    for k, f in enumerate(files):
        p = os.path.join(tmp, "c%d.br" % k)
        open(p, "wb").write(f)
        paths.append(p)
    r = subprocess.run([os.path.join(REF, "braid")] + paths, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    return r.returncode, r.stdout
byname_ = lambda n_: [c for c in cases if c[0] == n_][0][1]
for tag, names in (("skip-no-offsets", [nm("quickfox", ""), nm("x", "n"), nm("xyzzy", "u"), nm("10x10y", "")]),
                   ("only-skipped", [nm("x", "n")]),
                   ("none", []),
                   ("with-bare-trailer", [nm("quickfox", ""), "spec/shortest", nm("x", "")]),
                   ("bare-trailers", ["spec/shortest", "spec/shortest"]),
                   ("spec-shortest-data", ["spec/shortest-with-data", nm("x", "c1")])):
    rc_, out_ = braid_rc([byname_(n_) for n_ in names])
    add("braid/" + tag, out_)
    made["braid/" + tag] = {"braid": names, "ret": rc_}
for k, v in multis.items():
    if "braid" in made["multi/" + k]:  # braid's exit status: 1 when it skipped a file
        made["multi/" + k]["ret"] = braid_rc([byname_(n_) if n_ in [c[0] for c in cases] else multis[n_[6:]]
                                               for n_ in made["multi/" + k]["braid"]])[0]
    add("multi/" + k, v)
    add("multi/" + k + "+zeros", v + bytes(7))
    add("multi/" + k + "+garbage", v + b"\x55\xaa\x01")

# hand-made: intermediate headers without reverse offsets (a transmission-style file, only readable
# front to back), uncompressed lengths after the streams, a header check, an extra field and a method mask
def par(n):
    return (0x34cb00 >> ((n ^ (n >> 4)) & 0xf)) & 0x80
def var(n):
    out = bytearray()
    while n > 0x7f:
        out.append(n & 0x7f); n >>= 7
    out.append(0x80 | n)
    return bytes(out)
def bvar(n):
    out = bytearray([0x80 | (n & 0x7f)])
    n >>= 7
    while n > 0x7f:
        out.append(n & 0x7f); n >>= 7
    out.append(0x80 | n)
    return bytes(out)
o = H.oracle()
def chk(kind, data):
    if kind < 3:
        return (o.oracle_xxh32(data, len(data), 0) & ((1 << (8 << kind)) - 1)).to_bytes(1 << kind, "little")
    if kind == 3:
        return o.oracle_xxh64(data, len(data), 0).to_bytes(8, "little")
    return (o.oracle_crc32c(0, data, len(data)) & ((1 << (8 << (kind & 3))) - 1)).to_bytes(1 << (kind & 3), "little")
def segment(stream, plain, kind, with_len, extra=None, hdr_check=False, method=None, off=None):
    mask = kind | (8 if with_len else 0) | (0x10 if off is not None else 0)
    ex = b""
    if extra is not None or hdr_check or method is not None:
        mask |= 0x40
    head = bytearray()
    head.append(mask ^ par(mask))
    if off is not None:
        head += var(off)
    if mask & 0x40:
        e = (4 if extra is not None else 0) | (0x20 if hdr_check else 0) | (0x40 if method is not None else 0)
        head.append(e ^ par(e))
        if extra is not None:
            head += var(len(extra)) + extra
        if method is not None:
            head.append(method ^ par(method))
        if hdr_check:
            head += (o.oracle_xxh32(bytes(head), len(head), 0) & 0xffff).to_bytes(2, "little")
    return bytes(head) + stream + (var(len(plain)) if with_len else b"") + chk(kind, plain)
def plain(name):
    return open(os.path.join(TD, name), "rb").read()
def trailer(kind, off=None, total=None, checks=b""):
    mask = 0x20 | kind | (8 if total is not None else 0) | (0x10 if off is not None else 0)
    mask ^= par(mask)
    t = bytes([mask]) + (bvar(off) if off is not None else b"") + (bvar(total) if total is not None else b"")
    if kind != 7:
        t += chk(kind, checks)
    if kind != 7 or off is not None or total is not None:
        t += bytes([mask])
    return t
for tk in (7, 0, 1, 2, 3, 4, 5, 6):
    segs, checks, tot = [], b"", 0
    for k, s in enumerate(small):
        kind = k % 7
        sg = segment(td(s), plain(s), kind, with_len=(k % 2 == 0), extra=(b"\xc0\x81z" if k == 3 else None),
                     hdr_check=(k in (1, 3, 4)), method=(0 if k == 4 else None))
        segs.append(sg)
        checks += sg[-(1 << (kind & 3)):]
        tot += len(plain(s))
    add("stream/trailer-kind-%d" % tk, b"\xce\xb2\xcf\x81" + b"".join(segs) + trailer(tk, None, tot if tk % 2 else None, checks))

# a header that sits inside another header's extra field: a later header's reverse offset points at it, so a
# reader that indexes the file from the end sees it (with another kind of check) in front of the same stream,
# while the walk from the front goes through the real header and fails only at the later header's offset
def nested(kind_a, kind_in):
    seg0 = segment(td("quickfox"), plain("quickfox"), 0, with_len=True)
    A = 4 + len(seg0)
    mask_a = kind_a | 8 | 0x10 | 0x40
    head = bytearray([mask_a ^ par(mask_a)]) + var(A - 4)
    if kind_a == 7:
        head.append(0)
    head.append(4 ^ par(4))
    mask_in = kind_in | 8 | 0x10
    inner_at = A + len(head) + 1  # behind the one-byte length of the extra field
    inner = bytes([mask_in ^ par(mask_in)]) + var(inner_at - 4)
    assert len(inner) < 0x80
    head += var(len(inner)) + inner
    pa = plain("xyzzy")
    if kind_a == 7:
        import hashlib
        seg_a = bytes(head) + td("xyzzy") + var(len(pa)) + hashlib.sha256(pa).digest()
    else:
        seg_a = bytes(head) + td("xyzzy") + var(len(pa)) + chk(kind_a, pa)
    B = A + len(seg_a)
    seg_b = segment(td("x"), plain("x"), 0, with_len=True, off=B - inner_at)  # not B - A: the walk stops here
    T = B + len(seg_b)
    body = b"\xce\xb2\xcf\x81" + seg0 + seg_a + seg_b
    return body + trailer(0, T - B, len(plain("quickfox")) + len(pa) + len(plain("x")), b"")
for ka, ki in ((3, 4), (7, 3), (4, 2), (2, 7)):
    if ki == 7:
        continue  # an inner SHA-256 header needs its method byte: not this shape
    add("crafted/header-inside-extra-%d-%d" % (ka, ki), nested(ka, ki))

# damage: every byte of two small merged files under three masks, every truncation, and damage at
# random places of larger ones (most of it lands in compressed data)
rnd = random.Random(20260501)
for base in ("three-kinds", "eight-small"):
    v = multis[base]
    for pos in range(len(v)):
        for m in (0x01, 0x80, 0xff):
            if base == "eight-small" and m != 0x80 and pos % 3:
                continue
            b = bytearray(v)
            b[pos] ^= m
            add("damage/%s/%d^%02x" % (base, pos, m), bytes(b), ("multi/" + base, "xor", pos, m))
    for cut in range(len(v)):
        add("cut/%s/%d" % (base, cut), v[:cut], ("multi/" + base, "cut", cut, 0))
tk = [c for c in cases if c[0] == "stream/trailer-kind-2"][0][1]
for pos in range(len(tk)):
    b = bytearray(tk)
    m = 0x80 if pos % 2 else 0x04
    b[pos] ^= m
    add("damage/trailer-kind-2/%d" % pos, bytes(b), ("stream/trailer-kind-2", "xor", pos, m))
for base in ("text-12x64KiB", "large-mix", "mixed-16"):
    v = multis[base]
    for k in range(40):
        b = bytearray(v)
        pos, m = rnd.randrange(len(v)), 1 << rnd.randrange(8)
        b[pos] ^= m
        add("damage/%s/r%d" % (base, k), bytes(b), ("multi/" + base, "xor", pos, m))
    for k in range(10):
        cut = rnd.randrange(len(v))
        add("cut/%s/r%d" % (base, k), v[:cut], ("multi/" + base, "cut", cut, 0))

# cases whose verdict is undefined in the reference: a stream that runs past the end of the file is read
# from beyond the buffer there (broad.c:242 passes the length of the whole file), and a zero distance
# copies uninitialised memory.  Found with this repo's restatement; the tests do
# not compare those with the reference's verdict.
o.oracle_broad.restype = ctypes.c_int
over = ctypes.c_int.in_dll(o, "oracle_broad_overreads")
zero = ctypes.c_int.in_dll(o, "oracle_zero_distance_hits")
_libc_free = libc.free
_libc_free.argtypes = [ctypes.c_void_p]
def undefined_in_reference(data):
    a, b = over.value, zero.value
    out, n, seg = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
    o.oracle_broad(data, len(data), ctypes.byref(out), ctypes.byref(n), ctypes.byref(seg))
    _libc_free(out)
    return "overread" if over.value != a else "zero-distance" if zero.value != b else None

blob = bytearray()
index = []
for name, data, rc, out in cases:
    e = {"name": name, "rc": rc, "out_len": len(out), "out_xxh64": "%016x" % o.oracle_xxh64(out, len(out), 0)}
    u = undefined_in_reference(data)
    if u:
        e["reference_undefined"] = u
    if name in made:
        e["made"] = made[name]
    if name in recipes:
        e["base"], e["op"], e["pos"], e["mask"] = recipes[name]
    else:
        e["off"], e["len"] = len(blob), len(data)
        blob += data
    if len(out) <= 64:
        e["out"] = out.hex()
    index.append(e)
g = os.path.join(ROOT, "tests", "golden")
open(os.path.join(g, "br_streams.bin"), "wb").write(blob)
json.dump({"source": "tools/make_br_fixtures.py: files by the reference's brand/braid, verdicts by its broad()",
           "cases": index}, open(os.path.join(g, "br_streams.json"), "w"), indent=0)
import collections
print(len(cases), "cases,", len(blob), "bytes;", collections.Counter(c[2] for c in cases))

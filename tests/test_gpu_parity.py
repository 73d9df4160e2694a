"""GPU tier: the HIP path, called through the C-ABI, against the oracle and the golden fixtures.
Bar: bit-exact output, identical return code, identical got/used."""
import os
import random
import subprocess

import numpy as np
import pytest

import brotli_amd as ba
import helpers as H

pytestmark = pytest.mark.gpu


def gpu_yeast(src, expected=None):
    rc, out, got, used = ba.yeast(src, expected)
    return rc, got, used, out


def test_native_library_is_loaded_and_sees_a_gpu():
    assert os.path.exists(ba.LIB_PATH)
    assert ba.lib().yeast_device_count() >= 1


def test_crafted_good_and_bad_streams():
    cr = H.load_json("crafted.json")
    for name, v in cr.items():
        src = bytes.fromhex(v["hex"])
        rc, got, used, out = gpu_yeast(src)
        assert (rc, got, used, out.hex()) == (v["rc"], v["got"], v["used"], v["out"]), name
        assert (rc, got, used, out) == H.oracle_yeast(src), name


def test_testdata_corpus_decode_mode():
    idx = H.load_json("testdata.json")
    for name, comp, orig in H.testdata_pairs():
        rc, got, used, out = gpu_yeast(comp)
        assert (rc, got, used) == (0, idx[name]["got"], idx[name]["used"]), name
        assert out == orig, name


def test_testdata_corpus_compare_mode():
    # juxt.c:88 calls yeast(cmp=1); mismatch positions follow yeast.c:1217,1271,1287
    for name, comp, orig in H.testdata_pairs():
        assert gpu_yeast(comp, orig) == H.oracle_yeast(comp, orig), name
        if len(orig) < 4:
            continue
        rnd = random.Random(len(orig))
        for _ in range(4):
            e = bytearray(orig)
            e[rnd.randrange(len(e))] ^= 0x40
            e = bytes(e)
            assert gpu_yeast(comp, e) == H.oracle_yeast(comp, e), name
        for cut in (1, len(orig) // 2):
            assert gpu_yeast(comp, orig[:-cut]) == H.oracle_yeast(comp, orig[:-cut]), name
        assert gpu_yeast(comp, orig + b"xx") == H.oracle_yeast(comp, orig + b"xx"), name
    assert gpu_yeast(bytes.fromhex("06"), b"") == H.oracle_yeast(bytes.fromhex("06"), b"")


KERNELS = [64, 65, 1]  # cooperative, wavefront per stream (LDS tables), one stream per lane

# Every form of the decode path: the cooperative kernel, the wavefront-per-stream kernel, and the batch kernel
# (one stream per lane) in the three forms the dispatcher uses by batch size -- (YEAST_MERGE, YEAST_HALF) =
# (0, 1) plain loop, half-filled wavefronts; (2, 1) round 2's write-behind through the LDS ring, half-filled; (2, 0) the same in
# full wavefronts; (3, 1) the third loop form with 64-byte output chunks, half-filled (one round of a batch that fills the GPU:
# what bench.py runs at 65,536 streams); (3, 0) the same in full wavefronts (more than one round); (4, 1) and (4, 0) the third
# form with 16-byte chunks (what batches that do not fill the GPU get).
FORMS = [64, 65, (1, 0, 1), (1, 2, 1), (1, 2, 0), (1, 3, 1), (1, 3, 0), (1, 4, 1), (1, 4, 0)]
FORM_IDS = ["coop64", "wave", "lean-plain", "lean-ring-half", "lean-ring-full", "lean-kinds-half", "lean-kinds-full",
            "lean-kinds16-half", "lean-kinds16-full"]


def use_form(monkeypatch, form):
    """Force the kernel form; returns a small integer to seed per-form randomness with."""
    if isinstance(form, tuple):
        lanes, merge, half = form
        monkeypatch.setenv("YEAST_MERGE", str(merge))
        monkeypatch.setenv("YEAST_HALF", str(half))
        seed = 1 + 10 * merge + half
    else:
        lanes, seed = form, form
    monkeypatch.setenv("YEAST_FORCE_LANES", str(lanes))
    return seed


def test_compare_mode_against_the_reference_fixtures():
    # SURVEY 8c "cmp mode mismatch positions": (rc, got, used) of the real reference's yeast(cmp=1)
    from test_oracle_golden import cmp_cases
    n = 0
    for c, comp, expected in cmp_cases():
        rc, got, used, _ = gpu_yeast(comp, expected)
        assert (rc, got, used) == (c["rc"], c["got"], c["used"]), c
        n += 1
    assert n > 400


@pytest.mark.parametrize("lanes", KERNELS)
def test_crafted_streams_through_every_kernel(lanes, monkeypatch):
    monkeypatch.setenv("YEAST_FORCE_LANES", str(lanes))
    cr = H.load_json("crafted.json")
    names = sorted(cr)
    res = ba.yeast_batch([bytes.fromhex(cr[k]["hex"]) for k in names])
    for k, r in zip(names, res):
        v = cr[k]
        assert (r[0], r[2], r[3], r[1].hex()) == (v["rc"], v["got"], v["used"], v["out"]), k


@pytest.mark.parametrize("form", FORMS, ids=FORM_IDS)
def test_testdata_corpus_through_every_form(form, monkeypatch):
    # the reference's own corpus (real encoder output: several meta-blocks per file, ring-buffer and implicit distances
    # that reach across them, blocks that end in copies) as ONE batch through every form of the decode path -- the
    # default dispatch gives so few streams to the wavefront-per-stream kernel only
    use_form(monkeypatch, form)
    idx = H.load_json("testdata.json")
    pairs = list(H.testdata_pairs())
    res = ba.yeast_batch([comp for _, comp, _ in pairs])
    for (name, comp, orig), r in zip(pairs, res):
        assert (r[0], r[2], r[3]) == (0, idx[name]["got"], idx[name]["used"]), name
        assert r[1] == orig, name


@pytest.mark.parametrize("form", FORMS, ids=FORM_IDS)
def test_misc_encoder_parameters(form, monkeypatch):
    # window bits 10..24, qualities 0..10, NPOSTFIX/NDIRECT != 0 (unpinned by testdata/)
    use_form(monkeypatch, form)
    meta, streams = H.packed_streams("misc_streams")
    res = ba.yeast_batch(streams, ba.CHECK_CRC32C)
    for m, s, r in zip(meta["streams"], streams, res):
        rc, got, used, out = H.oracle_yeast(s)
        assert (r[0], r[2], r[3]) == (rc, got, used), m["params"]
        assert r[1] == out, m["params"]
        assert "%08x" % r[4] == m["crc32c"]


@pytest.mark.parametrize("form", FORMS, ids=FORM_IDS)
def test_synthesised_streams_cfg4(form, monkeypatch):
    # BASELINE config 4 (tools/brosynth.py; expected values from the reference): all 121 transforms at
    # every word length, the four context modes, NPOSTFIX/NDIRECT, block switching, context map
    # styles, simple codes, overlapping copies of every short period, stored and metadata blocks
    use_form(monkeypatch, form)
    meta, streams = H.packed_streams("cfg4_streams")
    res = ba.yeast_batch(streams, ba.CHECK_XXH64)
    for m, s, r in zip(meta["streams"], streams, res):
        assert (r[0], r[2], r[3]) == (m["rc"], m["got"], m["used"]), m["name"]
        assert "%016x" % r[4] == m["xxh64"], m["name"]
        assert r[1] == H.oracle_yeast(s)[3], m["name"]


@pytest.mark.parametrize("form", FORMS, ids=FORM_IDS)
def test_large_window_streams_cfg5(form, monkeypatch):
    # BASELINE config 5: window bits 24 with 16 MiB back-references and copies of megabytes (single
    # streams: a latency case).  The batch kernel (lanes 1) hands streams of more
    # than 1 MiB of output to its retry pass, a wavefront per stream.
    use_form(monkeypatch, form)
    meta, streams = H.packed_streams("cfg5_streams")
    for m, s in zip(meta["streams"], streams):
        (rc, out, got, used, chk), = ba.yeast_batch([s], ba.CHECK_CRC32C)
        assert (rc, got, used) == (m["rc"], m["got"], m["used"]), m["name"]
        assert "%08x" % chk == m["crc32c"], m["name"]
        assert "%016x" % ba.xxh64(out) == m["xxh64"], m["name"]


def test_large_streams_inside_a_large_batch():
    # 8,192 streams through the default dispatch (the batch kernel, one stream per lane), among them the 18 MB
    # stream with 16 MiB distances, the 2 MB one, and files of several meta-blocks (lcet10: 427 KB, alice29)
    meta, streams = H.packed_streams("cfg2_streams")
    m5, s5 = H.packed_streams("cfg5_streams")
    files = {}
    for name in ("lcet10.txt", "alice29.txt", "plrabn12.txt", "zeros"):
        files[name] = (open(os.path.join(H.GOLD, "testdata", name + ".compressed"), "rb").read(),
                       open(os.path.join(H.GOLD, "testdata", name), "rb").read())
    batch = list(streams) * 32
    want = [None] * len(batch)  # None: a cfg2 stream, checked by its fixture
    special = {}
    for k, (m, s) in enumerate(zip(m5["streams"], s5)):
        special[97 + 1000 * k] = (s, m)
    for k, name in enumerate(files):
        special[5 + 2040 * (k + 1)] = (files[name][0], files[name][1])
    for i, (s, w) in special.items():
        batch[i] = s
        want[i] = w
    assert len(batch) == 8192
    res = ba.yeast_batch(batch, ba.CHECK_XXH64)
    for i, r in enumerate(res):
        w = want[i]
        if w is None:
            m = meta["streams"][i % len(streams)]
            assert (r[0], r[2], r[3]) == (0, m["got"], m["len"]), i
            assert "%016x" % r[4] == m["xxh64"], i
        elif isinstance(w, dict):
            assert (r[0], r[2], r[3]) == (w["rc"], w["got"], w["used"]), w["name"]
            assert "%016x" % r[4] == w["xxh64"], w["name"]
        else:
            assert r[0] == 0 and r[1] == w and r[3] == len(batch[i]), i


@pytest.mark.parametrize("lanes", KERNELS)
def test_randomly_assembled_streams(lanes, monkeypatch):
    # 400 streams whose features are mixed by a seeded choice (tests/random_plans.py): several meta-blocks, context
    # modes and maps, NPOSTFIX/NDIRECT, block switching, dictionary words, ring and implicit distances, overlapping
    # and long copies; through every kernel, against the CPU restatement, bit for bit
    import random_plans as R
    monkeypatch.setenv("YEAST_FORCE_LANES", str(lanes))
    ss = R.streams(400, 20261111)
    res = ba.yeast_batch([s for s, _ in ss], ba.CHECK_CRC32C)
    o = H.oracle()
    for i, ((s, want), r) in enumerate(zip(ss, res)):
        rc, got, used, out = H.oracle_yeast(s)
        assert (rc, out) == (0, want), i
        assert (r[0], r[2], r[3]) == (rc, got, used), i
        assert r[1] == out, i
        assert r[4] == o.oracle_crc32c(0, out, len(out)), i
    # the same streams damaged (bit flips, cuts): same code, same progress, same partial output
    rnd = random.Random(5 + lanes)
    hurt = []
    for s, _ in ss[:250]:
        b = bytearray(s)
        if rnd.random() < 0.3:
            b = b[:rnd.randrange(1, len(b) + 1)]
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        hurt.append(bytes(b))
    for i, (s, r) in enumerate(zip(hurt, ba.yeast_batch(hurt))):
        rc, got, used, out = H.oracle_yeast(s)
        assert (r[0], r[2], r[3]) == (rc, got, used), i
        assert r[1] == out, i


@pytest.mark.parametrize("form", FORMS, ids=FORM_IDS)
def test_stream_found_by_the_soak(form, monkeypatch):
    # tools/soak_random.py, seed 3034, stream 770 (a damaged copy): a meta-block without a literal context map ends in a
    # copy, the next one has a map -- its first literal contexts are the bytes that copy wrote.  The cooperative kernel
    # kept p1/p2 only while a map was in use and decoded 803 bytes wrongly (the reference: rc 3, got 2218, used 202).
    # Beside it: intact streams of the same shape from the random plans (half of the blocks that another follows end
    # in a copy).
    import random_plans as R
    use_form(monkeypatch, form)
    s = open(os.path.join(H.GOLD, "soak_3034_770.bin"), "rb").read()
    cases = [s] + [x for x, _ in R.streams(120, 770)]
    want = [H.oracle_yeast(x) for x in cases]
    assert want[0][:3] == (3, 2218, 202)
    for i, (w, r) in enumerate(zip(want, ba.yeast_batch(cases))):
        assert (r[0], r[2], r[3]) == w[:3], i
        assert r[1] == w[3], i


@pytest.mark.parametrize("merge,half", [(1, 1), (2, 1), (2, 0), (3, 1), (3, 0), (4, 1), (4, 0)])
def test_batch_kernel_with_one_store_per_command(monkeypatch, merge, half):
    # the forms of the batch kernel that gather the stores of the output (YEAST_MERGE 1: a copy's last piece and the
    # literals behind it stored as one unit, the form large batches get; 2: every byte through a ring in LDS and out in
    # complete 16-byte chunks, near copies served from the ring): forced here
    # on small batches -- crafted streams, the corpus, random streams whole and damaged, outputs that end exactly at
    # the end of their region (the 16-byte store must not reach into the next stream's region)
    import random_plans as R
    monkeypatch.setenv("YEAST_FORCE_LANES", "1")
    monkeypatch.setenv("YEAST_MERGE", str(merge))
    monkeypatch.setenv("YEAST_HALF", str(half))  # 0: full wavefronts, what batches of more than one round get
    cr = H.load_json("crafted.json")
    cases = [bytes.fromhex(v["hex"]) for v in cr.values()]
    cases += [comp for _, comp, _ in H.testdata_pairs()]
    ss = R.streams(300, 99)
    cases += [s for s, _ in ss]
    rnd = random.Random(3)
    for s, _ in ss[:200]:
        b = bytearray(s)
        if rnd.random() < 0.3:
            b = b[:rnd.randrange(1, len(b) + 1)]
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        cases.append(bytes(b))
    want = [H.oracle_yeast(s) for s in cases]
    for i, (w, r) in enumerate(zip(want, ba.yeast_batch(cases, ba.CHECK_XXH64))):
        assert (r[0], r[2], r[3]) == w[:3], i
        assert r[1] == w[3], i
    # device call with regions exactly as long as the outputs, back to back: a byte written past a region shows up
    # as a wrong check of the next stream
    from brotli_amd.device import DeviceBatch
    good = [(s, w) for s, w in zip(cases, want) if w[0] == 0 and w[1] > 0][:400]
    b = DeviceBatch([s for s, _ in good], [w[1] for _, w in good], align=1)
    assert (np.diff(b.host_dst_off.astype(np.int64)) == np.array([w[1] for _, w in good])).all()
    b.decode(check=ba.CHECK_NONE, lanes=1)
    rec = b.records()
    for i, (s, w) in enumerate(good):
        assert (rec[i]["rc"], rec[i]["got"], rec[i]["used"]) == w[:3], i
    whole = bytes(b.dst[:int(b.host_dst_off[-1])].cpu().numpy())
    assert whole == b"".join(w[3] for _, w in good)


def test_host_batch_in_passes_under_a_memory_budget(monkeypatch):
    # a host batch larger than the share of device memory it may take goes in several passes with the same
    # results; a stream that does not fit the budget by itself is out of memory (code 1) and the others decode
    meta, streams = H.packed_streams("cfg2_streams")
    m5, s5 = H.packed_streams("cfg5_streams")
    monkeypatch.setenv("YEAST_HOST_BUDGET_MB", "8")
    batch = list(streams) + [s5[0]] + list(streams[:40])  # ~27 MB of output in all; s5[0] needs 18 MB
    res = ba.yeast_batch(batch, ba.CHECK_XXH64)
    for i, r in enumerate(res):
        if i == len(streams):
            assert r[0] == 1 and r[2] == 0, r[:1]
            continue
        m = meta["streams"][i if i < len(streams) else i - len(streams) - 1]
        assert (r[0], r[2], r[3]) == (0, m["got"], m["len"]), i
        assert "%016x" % r[4] == m["xxh64"], i
    monkeypatch.setenv("YEAST_HOST_BUDGET_MB", "32")
    r = ba.yeast_batch([s5[0]], ba.CHECK_XXH64)[0]
    assert (r[0], r[2]) == (0, m5["streams"][0]["got"]) and "%016x" % r[4] == m5["streams"][0]["xxh64"]


@pytest.mark.parametrize("lanes", [65, 64, 16, 8, 1, -1, -2, -3])
def test_cfg2_batch_every_group_width(lanes, monkeypatch):
    meta, streams = H.packed_streams("cfg2_streams")
    if lanes == -3:  # one stream per lane, streams handed out in batch order instead of by size class
        monkeypatch.setenv("YEAST_SORT", "0")
        lanes = 1
    if lanes < 0:  # one stream per lane in full (-1) or quarter-filled (-2) wavefronts; the default is half-filled
        monkeypatch.setenv("YEAST_HALF", "0" if lanes == -1 else "2")
        lanes = 1
    monkeypatch.setenv("YEAST_LANES", str(lanes))
    many = streams * 9  # 2,304 streams; YEAST_LANES names the kernel
    res = ba.yeast_batch(many, ba.CHECK_XXH64)
    for i, r in enumerate(res):
        m = meta["streams"][i % len(streams)]
        assert (r[0], r[2], r[3]) == (0, m["got"], m["len"]), i
        assert "%016x" % r[4] == m["xxh64"], i
    for i in (0, 100, 255, 300, 2303):
        assert res[i][1] == H.oracle_yeast(many[i])[3]


def test_check_values_known_answers():
    kat = H.load_json("checks.json")
    alice = open(os.path.join(H.GOLD, "testdata", "alice29.txt"), "rb").read()
    for k, v in kat.items():
        data = (b"123456789" if k == "123456789" else b"" if k == "empty" else alice if k == "alice29.txt"
                else alice[:int(k.split(":")[1][:-1])])
        assert "%08x" % ba.crc32c(data) == v["crc32c"], k
        assert "%08x" % ba.xxh32(data) == v["xxh32"], k
        assert "%016x" % ba.xxh64(data) == v["xxh64"], k
    o = H.oracle()
    a = ba.crc32c(alice[:1000])
    assert ba.crc32c(alice[1000:5000], a) == o.oracle_crc32c(0, alice[:5000], 5000)
    assert ba.xxh64(alice[:777], 12345) == o.oracle_xxh64(alice[:777], 777, 12345)
    assert ba.xxh32(alice[:777], 12345) == o.oracle_xxh32(alice[:777], 777, 12345)


def test_crc32c_of_long_buffers_and_of_few_streams():
    # a CRC-32C over one long buffer, or over a batch of few streams, is cut into 1,024 slices by a block of sixteen
    # wavefronts (64 slices by one wavefront otherwise): lengths around the slice and block boundaries, chained too
    o = H.oracle()
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, 5_000_001, dtype=np.uint8).tobytes()
    for n in (0, 1, 15, 16, 17, 1023, 1024, 16383, 16384, 16385, 65535, 65536, 65537, 65536 + 16 * 1024,
              (1 << 20) + 13, 5_000_001):
        assert ba.crc32c(data[:n]) == o.oracle_crc32c(0, data[:n], n), n
        assert ba.crc32c(data[:n], 0xdeadbeef) == o.oracle_crc32c(0xdeadbeef, data[:n], n), n
    # device-resident records of three streams (one of them empty): the block kernel
    from brotli_amd.device import DeviceBatch
    m5, s5 = H.packed_streams("cfg5_streams")
    streams = [s5[1], bytes.fromhex("06"), next(iter(H.testdata_pairs()))[1]]
    want = [H.oracle_yeast(x) for x in streams]
    b = DeviceBatch(streams, [max(w[1], 16) for w in want])
    b.decode(check=ba.CHECK_CRC32C)
    for r, w in zip(b.records(), want):
        assert (r["rc"], r["got"]) == (w[0], w[1])
        assert r["check"] == o.oracle_crc32c(0, w[3], len(w[3]))


@pytest.mark.parametrize("form", FORMS, ids=FORM_IDS)
def test_fuzzed_streams_same_error_class_and_progress(form, monkeypatch):
    # truncations and bit flips of real streams: rc, got, used and the partial output must agree
    rnd = random.Random(11 + use_form(monkeypatch, form))
    meta, streams = H.packed_streams("cfg2_streams")
    _, misc = H.packed_streams("misc_streams")
    cases = []
    for t in range(400):
        base = streams[t % 7] if t % 3 else misc[t % len(misc)]
        s = bytearray(base[:rnd.randrange(1, min(len(base), 6000))])
        if t % 2:
            for _ in range(rnd.randrange(1, 4)):
                s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
        cases.append(bytes(s))
    res = ba.yeast_batch(cases)
    classes = set()
    for s, r in zip(cases, res):
        rc, got, used, out = H.oracle_yeast(s)
        assert (r[0], r[2], r[3]) == (rc, got, used)
        assert r[1] == out
        classes.add(rc)
    assert {2, 3} <= classes


@pytest.mark.parametrize("form", [(1, 0, 1), (1, 2, 1), (1, 2, 0), (1, 3, 1), (1, 3, 0), 65, 16],
                         ids=["lean-plain", "lean-ring-half", "lean-ring-full", "lean-kinds-half", "lean-kinds-full", "wave", "coop16"])
def test_damage_deep_inside_full_length_streams(form, monkeypatch):
    # bit flips and cuts anywhere in whole 64 KiB streams: the failure (or the wrong but valid decode)
    # happens thousands of commands into the loop, beside healthy streams in the same wavefront
    rnd = random.Random(977 + use_form(monkeypatch, form))
    meta, streams = H.packed_streams("cfg2_streams")
    cases = []
    for t in range(768):
        s = bytearray(streams[rnd.randrange(len(streams))])
        kind = t % 4
        if kind == 0:
            s = s[:rnd.randrange(1, len(s))]
        elif kind < 3:
            for _ in range(kind):
                s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
        cases.append(bytes(s))  # kind 3: left whole
    res = ba.yeast_batch(cases, ba.CHECK_XXH64)
    classes = {}
    o = H.oracle()
    for k, (s, r) in enumerate(zip(cases, res)):
        rc, got, used, out = H.oracle_yeast(s)
        assert (r[0], r[2], r[3]) == (rc, got, used), k
        assert r[1] == out, k
        if rc == 0:
            assert r[4] == o.oracle_xxh64(out, len(out), 0), k
        classes[rc] = classes.get(rc, 0) + 1
    assert classes.get(0, 0) >= 192 and classes.get(2, 0) >= 100 and classes.get(3, 0) >= 100, classes


def test_full_size_batch_properties():
    # BASELINE config 2 at full size: 65,536 streams, checked through size-independent properties
    # (per-stream rc/got/used and a check-of-checks against the fixtures' reference values)
    import torch
    from brotli_amd.device import DeviceBatch
    meta, distinct = H.packed_streams("cfg2_streams")
    n = 65536
    rng = np.random.RandomState(3)
    order = rng.randint(0, len(distinct), size=n)
    batch = DeviceBatch([distinct[j] for j in order], [65536] * n)
    batch.decode(check=ba.CHECK_XXH64)
    torch.cuda.synchronize()
    r = batch.records()
    ms = meta["streams"]
    assert (r["rc"] == 0).all()
    assert (r["got"] == np.array([ms[j]["got"] for j in order], dtype=np.uint64)).all()
    assert (r["used"] == np.array([ms[j]["len"] for j in order], dtype=np.uint64)).all()
    assert (r["check"] == np.array([int(ms[j]["xxh64"], 16) for j in order], dtype=np.uint64)).all()
    for i in (0, 1, 4097, 65535):
        assert batch.output(i) == H.oracle_yeast(distinct[order[i]])[3]
    # second check kind over the same resident outputs, then a check of the checks
    batch.check(ba.CHECK_CRC32C)
    torch.cuda.synchronize()
    r = batch.records()
    want = np.array([int(ms[j]["crc32c"], 16) for j in order], dtype=np.uint64)
    assert (r["check"] == want).all()


@pytest.mark.parametrize("merge,half,n", [(2, 1, 72000), (2, 0, 72000), (3, 1, 72000), (3, 0, 72000), (4, 1, 72000), (3, 1, 65536)],
                         ids=["ring-half", "ring-full", "kinds-half", "kinds-full", "kinds16-half", "kinds-half-one-round"])
def test_wavefronts_take_further_streams_when_theirs_are_done(merge, half, n, monkeypatch):
    # The batch kernel holds 65,536 streams at a time (256 per CU); a wavefront whose streams are all done takes the
    # next S from the queue (yeast_lane.hip k_decode_lean's loop).  72,000 short streams -- random plans whole and
    # damaged, the crafted good and bad streams -- force that in both ring forms (S = 32 and S = 64); every record
    # and every output byte against the CPU restatement.  The reused wavefront inherits LDS (ring, maps, landing
    # slots) and workspace from the streams before, failed ones among them.  The last case is exactly one round (65,536
    # streams, every block resident): the static first hand-out that pairs long with short on a SIMD and the delayed
    # start of the wavefronts with shorter streams, on streams of every size, failed ones among them.
    import random_plans as R
    import torch
    from brotli_amd.device import DeviceBatch
    monkeypatch.setenv("YEAST_MERGE", str(merge))
    monkeypatch.setenv("YEAST_HALF", str(half))
    ss = [s for s, _ in R.streams(240, 4242 + half)]
    rnd = random.Random(17 + half)
    for s in ss[:120]:
        b = bytearray(s)
        if rnd.random() < 0.3:
            b = b[:rnd.randrange(1, len(b) + 1)]
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        ss.append(bytes(b))
    ss += [bytes.fromhex(v["hex"]) for v in H.load_json("crafted.json").values()]
    want = [H.oracle_yeast(s) for s in ss]
    o = H.oracle()
    chk = [o.oracle_xxh64(w[3], len(w[3]), 0) if w[0] == 0 else None for w in want]
    # output regions: a damaged stream may announce a meta-block far larger than what it gets to write, and the
    # device call reports a region smaller than the announcement with code 5 and the size wanted in `check`
    # (include/yeast_batch.h; the batch kernel asks at the meta-block header, before it has seen what is wrong with
    # the rest of the header) -- sized here by asking, streams that want over 256 KiB dropped
    need = [max(w[1], 1) + 40 for w in want]
    for _ in range(6):
        probe = DeviceBatch(ss, need, align=16)  # a region is as long as the distance to the next one
        probe.decode(check=ba.CHECK_NONE, lanes=1)
        pr = probe.records()
        small = np.nonzero(pr["rc"] == 5)[0]
        if len(small) == 0:
            break
        for j in small:
            need[j] = int(pr["check"][j]) + 40
    keep = [j for j in range(len(ss)) if need[j] <= (1 << 18) and pr["rc"][j] != 5]
    assert len(keep) > 300
    pick = np.array(keep)[np.random.RandomState(9 + half).randint(0, len(keep), size=n)]
    caps = [need[j] for j in pick]
    batch = DeviceBatch([ss[j] for j in pick], caps, align=16)
    batch.dst.fill_(0xEE)
    batch.decode(check=ba.CHECK_XXH64, lanes=1)
    torch.cuda.synchronize()
    assert ba.lib().yeast_last_kernel() == b"k_decode_lean<%d, %d>" % (32 if half else 64, merge)
    r = batch.records()
    w_rc = np.array([want[j][0] for j in pick])
    w_got = np.array([want[j][1] for j in pick], dtype=np.uint64)
    w_used = np.array([want[j][2] for j in pick], dtype=np.uint64)
    bad = np.nonzero((r["rc"] != w_rc) | (r["got"] != w_got) | (r["used"] != w_used))[0]
    assert len(bad) == 0, (bad[:5], r[bad[:5]], [want[pick[i]][:3] for i in bad[:5]])
    for i in np.nonzero(w_rc == 0)[0]:
        assert int(r["check"][i]) == chk[pick[i]], i
    out = batch.dst.cpu().numpy()
    off = batch.host_dst_off
    for i in range(n):  # partial outputs of failed streams too; and nothing written behind a stream's bytes
        j = pick[i]
        g = want[j][1]
        a = int(off[i])
        assert out[a:a + g].tobytes() == want[j][3], i
        assert (out[a + g:int(off[i + 1])] == 0xEE).all() or want[j][0] != 0, i


def test_shard_count_invariance():
    # the batch cut into N contiguous shards balanced by compressed bytes gives, shard
    # after shard, exactly the records and the bytes of the uncut batch (every stream is independent:
    # yeast.c:1315-1336 resets all state per call)
    import torch
    from brotli_amd.device import DeviceBatch
    from brotli_amd.sharding import balanced_shards
    meta, distinct = H.packed_streams("cfg2_streams")
    n = 16384
    order = np.random.RandomState(5).randint(0, len(distinct), size=n)
    streams = [distinct[j] for j in order]
    whole = DeviceBatch(streams, [65536] * n)
    whole.decode(check=ba.CHECK_XXH64)
    torch.cuda.synchronize()
    want = whole.records()
    assert (want["rc"] == 0).all()
    want_bytes = whole.dst[:n * 65536].cpu().numpy()
    for shards in (2, 4, 8):
        cuts = balanced_shards([len(s) for s in streams], shards)
        recs, outs = [], []
        for lo, hi in cuts:
            part = DeviceBatch(streams[lo:hi], [65536] * (hi - lo))
            part.decode(check=ba.CHECK_XXH64)
            torch.cuda.synchronize()
            recs.append(part.records())
            outs.append(part.dst[:(hi - lo) * 65536].cpu().numpy())
        assert np.concatenate(recs).tobytes() == want.tobytes(), shards
        assert np.array_equal(np.concatenate(outs), want_bytes), shards


def test_two_batches_in_flight_on_two_streams():
    # every call on a device shares the library's counters and table workspaces: a batch enqueued on another
    # stream must wait for the one before it (include/yeast_batch.h)
    import torch
    from brotli_amd.device import DeviceBatch
    meta, distinct = H.packed_streams("cfg2_streams")
    a = DeviceBatch([distinct[j % 256] for j in range(9000)], [65536] * 9000)
    b = DeviceBatch([distinct[(7 * j + 3) % 256] for j in range(8500)], [65536] * 8500)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        with torch.cuda.stream(s1):
            a.decode(check=ba.CHECK_XXH64)
        with torch.cuda.stream(s2):
            b.decode(check=ba.CHECK_XXH64)
    torch.cuda.synchronize()
    for batch, f in ((a, lambda j: j % 256), (b, lambda j: (7 * j + 3) % 256)):
        r = batch.records()
        assert (r["rc"] == 0).all() and (r["got"] == 65536).all()
        for j in range(batch.n):
            assert "%016x" % int(r["check"][j]) == meta["streams"][f(j)]["xxh64"], j


def test_entry_points_from_many_host_threads():
    # SURVEY 8b Threading: yeast() keeps no global state (try.h:353-361), so callers may call it from many
    # threads at once; here eight threads call yeast(), yeast_batch() and the check functions together
    import threading
    meta, distinct = H.packed_streams("cfg2_streams")
    idx = H.load_json("testdata.json")
    pairs = list(H.testdata_pairs())
    errors = []

    def work(t):
        try:
            rnd = random.Random(t)
            for rep in range(6):
                name, comp, orig = pairs[rnd.randrange(len(pairs))]
                rc, out, got, used = ba.yeast(comp)
                assert (rc, out) == (0, orig), name
                rc, _, got, used = ba.yeast(comp, orig)
                assert (rc, got, used) == (0, len(orig), idx[name]["used"]), name
                pick = [rnd.randrange(256) for _ in range(40)]
                res = ba.yeast_batch([distinct[j] for j in pick], ba.CHECK_CRC32C)
                for j, r in zip(pick, res):
                    assert (r[0], r[2]) == (0, 65536) and "%08x" % r[4] == meta["streams"][j]["crc32c"]
                assert "%016x" % ba.xxh64(orig) == idx[name]["xxh64"] if "xxh64" in idx[name] else True
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert ba.lib().yeast_check_error() == 0


def test_device_api_reports_small_output_region():
    from brotli_amd.device import DeviceBatch
    import torch
    meta, distinct = H.packed_streams("cfg2_streams")
    batch = DeviceBatch(distinct[:4], [65536, 1000, 65536, 65472])
    batch.decode(check=0, lanes=64)
    torch.cuda.synchronize()
    r = batch.records()
    assert list(r["rc"]) == [0, 5, 0, 5]
    assert r["check"][1] == 65536 and r["got"][1] == 0


def test_trim_gives_buffers_back_and_the_next_calls_allocate_again():
    meta, distinct = H.packed_streams("cfg2_streams")
    # enough bytes for pinned staging in both directions and for the host threads that pack and cut up
    streams = [distinct[k % len(distinct)] for k in range(600)]
    first = ba.yeast_batch(streams, ba.CHECK_XXH64)
    for k, (rc, data, got, used, chk) in enumerate(first):
        m = meta["streams"][k % len(distinct)]
        assert (rc, got, used, chk) == (0, m["got"], m["len"], int(m["xxh64"], 16)), k
        assert ba.xxh64(data) == chk if k < 4 else len(data) == got
    ba.trim()
    ba.trim()  # nothing held: still fine
    assert ba.yeast_batch(streams, ba.CHECK_XXH64) == first
    src = distinct[0]
    assert gpu_yeast(src) == H.oracle_yeast(src)
    assert ba.crc32c(b"123456789") == 0xe3069283  # the scratch buffer of the check calls comes back too


def test_reference_harnesses_linked_against_the_drop_in():
    # BASELINE config 1: the reference's own juxt.c/deb.c, compiled unmodified in the build
    # container and linked against libyeast_mi355x.so (oracle/Makefile `dropin`)
    juxt = os.path.join(H.ROOT, "oracle", "_ref", "juxt_dropin")
    if not os.path.exists(juxt):
        pytest.skip("drop-in harness binaries were not built")
    files = [os.path.join(H.GOLD, "testdata", f) for f in sorted(os.listdir(os.path.join(H.GOLD, "testdata")))
             if f.endswith(".compressed")]
    p = subprocess.run([juxt, "-v"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0
    assert b"yeast() returned" not in p.stderr, p.stderr[-400:]
    assert p.stderr.count(b":\n") == 17


def test_br_container_reader_every_fixture():
    # broad() (broad.c:142-353) on the batch decoder.  Every case against this repo's
    # restatement (code, bytes written, streams decoded) and, where the reference's behaviour is
    # defined, against the reference's own verdict; hand-made, merged, damaged and truncated files.
    import hashlib
    cases = H.br_cases()
    for k, (c, data) in enumerate(cases):
        big = c["out_len"] > 100000 or len(data) > 100000
        if c["name"].startswith(("damage/", "cut/")) and (k % 3 and not big):
            continue  # a third of the small damaged files, every large one
        got = ba.broad(data)
        assert got == H.oracle_broad(data), c["name"]
        if not c.get("reference_undefined"):
            assert (got[0], len(got[1])) == (c["rc"], c["out_len"]), c["name"]
            assert "%016x" % ba.xxh64(got[1]) == c["out_xxh64"], c["name"]
    for msg in (b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 64, bytes(range(256)) * 999):
        assert ba.sha256(msg) == hashlib.sha256(msg).digest()
    c, data = [x for x in cases if x[0]["name"] == "multi/text-12x64KiB"][0]
    assert ba.broad(data, write=False) == (0, b"", 12)
    # a header inside another header's extra field, found by the index from the end with another kind of check in
    # front of the same stream: the decode-ahead result must not be used for the real header (verifying only, too)
    nested = [x for x in cases if x[0]["name"].startswith("crafted/header-inside-extra")]
    assert len(nested) == 3
    for c, data in nested:
        assert ba.broad(data, write=False) == (c["rc"], b"", 2), c["name"]


def test_br_writer_and_merger_byte_exact():
    # brand (brand.c:93-314) and braid (braid.c:146-422).  Expected bytes are the output of
    # the reference's own programs on the same inputs (tools/make_br_fixtures.py); merges include files
    # braid skips (no chain of reverse offsets) with its exit status, bare trailers and the empty merge.
    cases = H.br_cases()
    by = {c["name"]: d for c, d in cases}
    meta2, cfg2 = H.packed_streams("cfg2_streams")

    def td(name):
        return open(os.path.join(H.GOLD, "testdata", name + ".compressed"), "rb").read()
    nb = nm = 0
    for c, data in cases:
        m = c.get("made")
        if not m:
            continue
        if "brand" in m:
            s, opt, name = m["brand"]
            assert ba.brand(td(s), opt, name) == (0, data), c["name"]
            nb += 1
        elif "braid" in m:
            assert ba.braid([by[n] for n in m["braid"]]) == (m.get("ret", 0), data), c["name"]
            nm += 1
        else:
            parts = []
            for kind, s, opt in m["braid_brand"]:
                rc, b = ba.brand(cfg2[s] if kind == "cfg2" else td(s), opt)
                assert rc == 0
                parts.append(b)
            assert ba.braid(parts) == (0, data), c["name"]
            nm += 1
    assert nb > 60 and nm >= 14
    # a stream with trailing bytes, or a damaged one, is refused as brand refuses it (brand.c:300-304)
    assert ba.brand(td("quickfox") + b"\0")[0] == 1
    assert ba.brand(td("quickfox")[:-3])[0] == 1
    # what is written here reads back through the reader
    rc, b = ba.brand(cfg2[3], "c2u")
    assert ba.broad(b) == (0, H.oracle_yeast(cfg2[3])[3], 1)


def test_merger_survives_wild_reverse_offsets():
    # braid.c:182,195 subtract whatever distance the file gives from an off_t: 2^63 and more moves the offset
    # forward, more than the offset makes it negative.  Reading there is end of file (status 1, file skipped)
    # or "invalid distance" in the reference; here it must never leave the buffer.
    def par(n):
        return (0x34cb00 >> ((n ^ (n >> 4)) & 0xf)) & 0x80

    def bvar(n):
        out = [0x80 | (n & 0x7f)]
        n >>= 7
        while n > 0x7f:
            out.append(n & 0x7f)
            n >>= 7
        out.append(0x80 | n)
        return bytes(out)

    def var(n):
        out = []
        while n > 0x7f:
            out.append(n & 0x7f)
            n >>= 7
        out.append(0x80 | n)
        return bytes(out)
    sig = bytes([0xce, 0xb2, 0xcf, 0x81])
    trail = 0x20 | 0x10 | 7
    trail ^= par(trail)
    rc, good = ba.brand(open(os.path.join(H.GOLD, "testdata", "quickfox.compressed"), "rb").read(), "")
    assert rc == 0
    empty = ba.braid([])
    wild = []
    for dist in ((1 << 63) + 5, (1 << 64) - 1, 10 ** 6, (1 << 63), 3):
        wild.append(sig + bytes(40) + bytes([trail]) + bvar(dist) + bytes([trail]))
    # a header whose own reverse offset is wild: header mask with an offset, reached from a sane trailer
    hm = 0x10 | 7
    hm ^= par(hm)
    for back in ((1 << 63) + 1, (1 << 64) - 3, 1 << 40):
        body = sig + bytes(8) + bytes([hm]) + var(back) + bytes([0]) + bytes(20)
        dist = len(body) - 12
        wild.append(body + bytes([trail]) + bvar(dist) + bytes([trail]))
    for w in wild:
        assert ba.braid([w]) == (1, empty[1]), w.hex()
        assert ba.braid([good, w, good])[0] == 1
        assert ba.braid([good, w, good])[1] == ba.braid([good, good])[1]
        assert ba.broad(w)[0] != 0


def test_streaming_hashes_any_chunking():
    # XXH32/XXH64 reset/update/digest (xxhash.c:588-960) and chained crc32c (crc32c.c:258):
    # whatever the chunking, the digest is the one-shot value of the restatement; digests in mid-stream
    # do not disturb the state; the check of checks of a merged file is reproduced the way braid builds it
    o = H.oracle()
    rnd = random.Random(5)
    alice = open(os.path.join(H.GOLD, "testdata", "alice29.txt"), "rb").read()
    for trial in range(24):
        data = alice[rnd.randrange(2000):][:rnd.choice([0, 1, 15, 16, 17, 31, 32, 33, 100, 4096, 50001])]
        seed = rnd.choice([0, 1, 0x9E3779B1])
        s32, s64, crc, at = ba.XXHState(32, seed), ba.XXHState(64, seed), 0, 0
        while at < len(data):
            n = rnd.choice([0, 1, 3, 15, 16, 17, 31, 32, 33, 64, 1000, 7777])
            piece = data[at:at + n]
            s32.update(piece)
            s64.update(piece)
            crc = ba.crc32c(piece, crc)
            at += len(piece)
            if rnd.random() < 0.3:
                assert s32.digest() == o.oracle_xxh32(data[:at], at, seed)
                assert s64.digest() == o.oracle_xxh64(data[:at], at, seed)
        assert s32.digest() == o.oracle_xxh32(data, len(data), seed), (trial, len(data))
        assert s64.digest() == o.oracle_xxh64(data, len(data), seed), (trial, len(data))
        assert crc == o.oracle_crc32c(0, data, len(data))
    # the XXH32 check of checks in the trailer of a merged file, built as braid builds it (braid.c:337,414):
    # one update per segment with that segment's stored check value (here XXH64, eight bytes each)
    c, data = [x for x in H.br_cases() if x[0]["name"] == "multi/eight-small"][0]
    st = ba.XXHState(32, 0)
    for name in c["made"]["braid"]:
        plain = open(os.path.join(H.GOLD, "testdata", name.split("/")[1]), "rb").read()
        st.update(o.oracle_xxh64(plain, len(plain), 0).to_bytes(8, "little"))
    assert st.digest() == int.from_bytes(data[-5:-1], "little")


@pytest.mark.parametrize("lanes", KERNELS)
def test_streams_written_in_the_brogen_notation(lanes, monkeypatch):
    # complex prefix codes described with tools/brogen.py's `c` command (every length 1..15, long
    # repeats of a length and of zero, a two-symbol code, a stream cut short inside its literals)
    monkeypatch.setenv("YEAST_FORCE_LANES", str(lanes))
    own = H.own_streams()
    names = sorted(own)
    res = ba.yeast_batch([own[k][0] for k in names])
    for k, r in zip(names, res):
        rc, got, used, out = H.oracle_yeast(own[k][0])
        assert (r[0], r[2], r[3], r[1]) == (rc, got, used, out), k
        if own[k][1] is not None:
            assert (r[0], r[1]) == (0, own[k][1]), k


def test_reference_br_programs_linked_against_the_drop_in(tmp_path):
    # the reference's own broad.c, brand.c and braid.c, compiled unmodified and linked against
    # libyeast_mi355x.so for yeast(), crc32c(), XXH32(), XXH64() and the streaming XXH calls
    # (oracle/Makefile `brdropin`; SHA-256 from the image's libcrypto, as in the reference)
    ref = os.path.join(H.ROOT, "oracle", "_ref")
    if not all(os.path.exists(os.path.join(ref, b)) for b in ("broad_dropin", "brand_dropin", "braid_dropin")):
        pytest.skip("drop-in .br binaries were not built")
    cases = H.br_cases()
    by = {c["name"]: (c, d) for c, d in cases}

    def td(name):
        return open(os.path.join(H.GOLD, "testdata", name + ".compressed"), "rb").read()
    # brand: the same bytes as the fixtures (which the reference's own build wrote)
    for name in ("single/quickfox/c2", "single/x/s", "single/xyzzy/x4f", "single/alice29.txt/default"):
        c, want = by[name]
        s, opt, fname = c["made"]["brand"]
        args = [os.path.join(ref, "brand_dropin")] + ([opt] if opt or fname else []) + ([fname] if fname else [])
        p = subprocess.run(args, input=td(s), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.stdout == want, name
    # braid over files on disk
    parts = []
    for k, name in enumerate(by["multi/three-kinds"][0]["made"]["braid"]):
        f = tmp_path / ("p%d.br" % k)
        f.write_bytes(by[name][1])
        parts.append(str(f))
    p = subprocess.run([os.path.join(ref, "braid_dropin")] + parts, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.stdout == by["multi/three-kinds"][1]
    # broad: data written and the verdict (it prints "broad() error: ..." and nothing else on failure)
    for name in ("multi/three-kinds", "multi/mixed-16", "multi/large-mix", "stream/trailer-kind-3", "spec/shortest"):
        c, data = by[name]
        p = subprocess.run([os.path.join(ref, "broad_dropin")], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.stderr == b"" and len(p.stdout) == c["out_len"], (name, p.stderr[-200:])
        assert "%016x" % ba.xxh64(p.stdout) == c["out_xxh64"], name
    c, data = by["damage/three-kinds/60^ff"] if "damage/three-kinds/60^ff" in by else by["cut/three-kinds/70"]
    p = subprocess.run([os.path.join(ref, "broad_dropin")], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert (p.stderr != b"") == (c["rc"] != 0)

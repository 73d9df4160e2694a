#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: decompressed GB/s (whole node), batched 64 KiB brotli streams.

One step = one pass of the hot path over one batch: decode every stream of the batch
(yeast_batch_device), compute the XXH64 check of every output (fused into the same call), and --
for N > 1 -- all-gather the 32-byte result records over RCCL.  Inputs are resident in HBM before
the timed region starts.

Workload (config.workload = "cfg2"): 65,536 independent streams, each decoding to 65,536 bytes (English
text, brotli quality 11, lgwin 22), tiled from the 256 distinct committed streams of
tests/golden/cfg2_streams.bin in a fixed pseudo-random order.

N = 1: BASELINE.json config 2, the whole batch on one GPU ("scaling": "weak" trivially).
N > 1: BASELINE.json config 3, "same batch sharded" -- the default is --scaling strong: ONE batch of
--streams streams for the whole job, cut into contiguous shards by compressed bytes
(brotli_amd.sharding.balanced_shards), one per rank; the records are all-gathered inside the step,
value = bytes of the whole batch / max-over-ranks time.  The line then carries "weak_scaling" (every
rank its own --streams batch, from a second timed loop) and "strong_scaling_floor": the single-GPU
kernel time over batch sizes (profiles/r0N/stream_series.json), i.e. what a shard of that size costs
-- a launch lasts as long as its longest stream, so a shard of 8,192 streams is not 8 x faster than
the batch of 65,536.  --scaling weak asks for the weak-scaling run alone.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 through
python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def load_distinct():
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg2_streams.json")))
    blob = open(os.path.join(ROOT, "tests", "golden", "cfg2_streams.bin"), "rb").read()
    streams = [blob[m["off"]:m["off"] + m["len"]] for m in meta["streams"]]
    return meta["streams"], streams


def build_order(n, ndistinct, seed, kind="shuffled"):
    """stream i of the batch = distinct[order[i]]: every distinct stream equally often, order permuted.
    kind "wave" (64 consecutive copies of one stream) and "same" (one stream only) are diagnostic
    layouts that remove lane divergence; the bench line is always quoted on "shuffled"."""
    if kind == "wave":
        return (np.arange(n) // 64 * 7) % ndistinct
    if kind == "same":
        return np.zeros(n, dtype=np.int64)
    rng = np.random.RandomState(seed)
    reps = (n + ndistinct - 1) // ndistinct
    order = np.tile(np.arange(ndistinct), reps)[:n]
    rng.shuffle(order)
    return order


def cpu_baseline(streams, order, cores, sample):
    """Time the reference decoder (oracle/_ref, built from /root/reference in the build container)
    -- or, if absent, this repo's CPU restatement -- on a bounded sample of the same batch."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    import helpers as H
    ref = H.reference()
    kind = "reference" if ref is not None else "port"
    fn = ref.yeast if ref is not None else H.oracle().oracle_yeast
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    bufs = [ctypes.create_string_buffer(s, len(s)) for s in streams]
    idx = [int(order[i % len(order)]) for i in range(sample)]

    def work(lo, hi):
        tot = 0
        for j in idx[lo:hi]:
            dest = ctypes.c_void_p(None)
            got = ctypes.c_size_t(0)
            ln = ctypes.c_size_t(len(streams[j]))
            rc = fn(ctypes.byref(dest), ctypes.byref(got), bufs[j], ctypes.byref(ln), 0)
            assert rc == 0
            tot += got.value
            libc.free(dest)
        return tot
    per = (sample + cores - 1) // cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        futs = [ex.submit(work, k * per, min(sample, (k + 1) * per)) for k in range(cores)]
        total = sum(f.result() for f in futs)
    dt = time.perf_counter() - t0
    return {"value": total / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": "%d streams of the cfg2 batch (%d bytes out), one yeast() call per stream over %d threads"
                      % (sample, total, cores)}


def kernel_sources_hash():
    """sha256 over the kernel sources: figures measured on another version of the kernels (profiles/hbm_traffic.json,
    profiles/r03/ceilings.json carry the hash they were measured on) are not quoted as this run's."""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "brotli_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "brotli_amd", "csrc", "*.hip"))):
        text = open(f, "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)  # comments and layout do not change the kernels
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(os.path.basename(f).encode())
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def measured_copy_gbs(torch, dev, nbytes=1 << 30, reps=5):
    """HBM-to-HBM copy rate on this GPU (bytes read + bytes written per second), the practical
    ceiling beside the 8 TB/s spec figure (SURVEY 8d side figure)."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (default: the cfg2 batch)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of a wavefront per stream (0 = library default)")
    ap.add_argument("--check", type=int, default=3, help="0 none, 1 crc32c, 2 xxh32, 3 xxh64")
    ap.add_argument("--order", default="shuffled", choices=["shuffled", "wave", "same"],
                    help="diagnostic stream layouts (see build_order); the metric is quoted on shuffled")
    ap.add_argument("--cpu-sample", type=int, default=16384, help="streams for the host baseline (0 = skip)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --streams per GPU; strong: --streams for the whole job, sharded over the ranks "
                         "(BASELINE config 3).  Default: strong when launched on more than one GPU, else weak")
    ap.add_argument("--no-weak", action="store_true", help="N > 1, strong scaling: skip the second (weak-scaling) timed loop")
    args = ap.parse_args()

    import torch
    import brotli_amd as ba
    from brotli_amd.device import DeviceBatch, RECORD_DTYPE
    from brotli_amd.sharding import all_gather_records

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    rehearsal = bool(os.environ.get("BENCH_TEST_BACKEND")) and world > 1
    if world > 1:
        import torch.distributed as dist
        # BENCH_TEST_BACKEND=gloo: a rehearsal of the N > 1 code path on a box with ONE GPU (every rank on device 0, the
        # records gathered through gloo); never a measurement
        test_backend = os.environ.get("BENCH_TEST_BACKEND")
        if test_backend:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(test_backend)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    meta, distinct = load_distinct()
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    strong = scaling == "strong"
    verified = [True]

    def barrier():
        if dist is not None:
            dist.barrier()

    def prepare(strong_mode):
        """The batch of one mode, resident in HBM, with what the fixtures say about every stream of it."""
        full = None
        if strong_mode:
            # one batch for the whole job (the same on every rank), this rank's contiguous shard of it
            from brotli_amd.sharding import balanced_shards
            full = build_order(args.streams, len(distinct), 0x5EED, args.order)
            shards = balanced_shards([len(distinct[j]) for j in full], world)
            lo, hi = shards[rank]
            order = full[lo:hi]
            sizes = [(b - a) * 32 for a, b in shards]
            if hi == lo:
                raise SystemExit("bench: fewer streams than ranks")
        else:
            order = build_order(args.streams, len(distinct), 0x5EED + rank, args.order)
            sizes = [len(order) * 32] * world  # every rank decodes as many streams: no exchange of counts in the step
        n = len(order)
        batch = DeviceBatch([distinct[j] for j in order], [65536] * n, device=dev)
        want = {"got": np.array([meta[j]["got"] for j in order], dtype=np.uint64),
                "used": np.array([meta[j]["len"] for j in order], dtype=np.uint64),
                "chk": np.array([int(meta[j][{1: "crc32c", 2: "xxh32", 3: "xxh64"}.get(args.check, "xxh64")], 16)
                                 for j in order], dtype=np.uint64)}
        return {"batch": batch, "order": order, "full": full, "sizes": sizes, "n": n, "want": want}

    def timed(m):
        """W warm-up steps, the fixture check outside the timed region, K timed steps: seconds (max over ranks)."""
        batch, sizes, want = m["batch"], m["sizes"], m["want"]

        def step():
            batch.decode(check=args.check, lanes=args.lanes)
            if dist is not None:
                return all_gather_records(batch.rec.cpu() if rehearsal else batch.rec, dist, sizes=sizes)
            return batch.rec
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        m["kernel"] = ba.lib().yeast_last_kernel().decode()  # what the library's dispatcher launched for this batch
        # verify once, outside the timed region: every stream decoded, sizes and checks as the fixtures say
        r = batch.records()
        ok = (r["rc"] == 0).all() and (r["got"] == want["got"]).all() and (r["used"] == want["used"]).all()
        if args.check:
            ok = ok and (r["check"] == want["chk"]).all()
        if not ok:
            verified[0] = False
            if not os.environ.get("BENCH_NOVERIFY"):  # BENCH_NOVERIFY: timing of deliberately broken experiment builds
                bad = int(np.argmax((r["rc"] != 0) | (r["got"] != want["got"]) | (r["used"] != want["used"]) |
                                    ((r["check"] != want["chk"]) & bool(args.check))))
                raise SystemExit("bench: decode results do not match the fixtures (first bad stream %d: %s)" % (bad, r[bad]))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    m = prepare(strong)
    dt = timed(m)
    batch, n, order, full = m["batch"], m["n"], m["order"], m["full"]
    want_got, want_used = m["want"]["got"], m["want"]["used"]

    # kernel-level timing for the roofline: HIP events on the stream the kernels are launched on,
    # decode only (no check, no gather), K launches
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        batch.decode(check=0, lanes=args.lanes)
        b.record()
    torch.cuda.synchronize()
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    evc = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if args.check:
        for a, b in evc:
            a.record()
            batch.check(args.check)
            b.record()
        torch.cuda.synchronize()
        c_ms = float(np.mean([a.elapsed_time(b) for a, b in evc]))
    else:
        c_ms = 0.0

    out_bytes = float(want_got.sum())
    in_bytes = float(want_used.sum())
    algo = out_bytes + in_bytes  # B = used_in + got_out per stream
    ms_per_step = dt / args.steps * 1e3
    if strong:  # the whole batch's bytes: every rank knows the batch
        job_out = float(sum(meta[j]["got"] for j in full))
    else:
        job_out = out_bytes * world
    value = job_out / (dt / args.steps) / 1e9
    copy_gbs = measured_copy_gbs(torch, dev) if rank == 0 else None
    achieved = algo / (k_ms * 1e-3) / 1e9
    kernel = m["kernel"]
    # HBM bytes per launch from the PMC passes (tools/prof_summary.py writes the file).  Only quoted when it was
    # measured on this batch size AND on the kernel that ran just now: a figure of another kernel is no figure.
    traffic, traffic_note = None, None
    src_hash = kernel_sources_hash()
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        try:
            t = json.load(open(tf))
            if t.get("streams") == n and kernel and kernel in str(t.get("kernel", "")) and t.get("csrc") != src_hash:
                traffic_note = "profiles/hbm_traffic.json was measured on kernel sources %s, these are %s: not quoted" % (t.get("csrc"), src_hash)
            elif t.get("streams") == n and kernel and kernel in str(t.get("kernel", "")):
                traffic = t.get("bytes_per_launch")
                traffic_note = {k: t[k] for k in ("fetch_bytes", "write_bytes", "fetch_factor", "source") if k in t}
            elif t.get("streams") == n:
                traffic_note = "profiles/hbm_traffic.json is of kernel %r, this run launched %r: not quoted" % (t.get("kernel"), kernel)
        except Exception:
            traffic = None

    # the launch against the three ceilings that have been measured for it: HBM bytes, the rate at
    # which HBM serves unrelated 64-byte lines, the instruction issue rate of one wavefront over the longest stream.
    # Each: the time the ceiling alone would allow, and that time as a fraction of the kernel time measured just now.
    ceilings = {"hbm_bytes": {"bound_ms": algo / (HBM_PEAK_GBS * 1e9) * 1e3,
                              "frac": algo / (HBM_PEAK_GBS * 1e9) * 1e3 / k_ms}}
    cf = os.path.join(ROOT, "profiles", "r03", "ceilings.json")
    if os.path.exists(cf):
        try:
            c = json.load(open(cf))
            if c.get("streams") == n and kernel and kernel in str(c.get("kernel", "")) and c.get("csrc") == src_hash:
                hl, it = c["hbm_lines"], c["issue"]
                b1 = hl["lines_per_launch"] / hl["ceiling_lines_per_s"] * 1e3
                b2 = it["longest_stream_trips"] * it["instructions_per_trip"] * it["cycles_per_instruction"] / (it["clock_ghz"] * 1e9) * 1e3
                ceilings["hbm_lines"] = {"bound_ms": b1, "frac": b1 / k_ms, "lines_per_launch": hl["lines_per_launch"],
                                         "ceiling_lines_per_s": hl["ceiling_lines_per_s"]}
                ceilings["wavefront_issue"] = {"bound_ms": b2, "frac": b2 / k_ms,
                                               **{k: it[k] for k in ("longest_stream_trips", "instructions_per_trip",
                                                                     "cycles_per_instruction", "clock_ghz")}}
                ceilings["source"] = "profiles/r03/ceilings.json (counters and listing of %s at %d streams)" % (c["kernel"], n)
        except Exception:
            pass

    line = {
        "metric": "decompressed GB/s (whole node), batched 64 KiB brotli streams",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "verified": verified[0],
        "dtype": "u8", "data": "synthetic" if not os.environ.get("BENCH_TEST_BACKEND") else "REHEARSAL: all ranks on one GPU, not a measurement",
        "config": {"workload": "cfg2: %s x 65,536 B out, q11 lgwin22 English text, %d distinct "
                               "streams tiled in seeded order; decode + %s check%s"
                               % ("%d streams in all, sharded by compressed bytes over %d GPU(s)" % (args.streams, world)
                                  if strong else "%d streams/GPU" % n,
                                  len(distinct), {0: "no", 1: "CRC-32C", 2: "XXH32", 3: "XXH64"}[args.check],
                                  " + RCCL all-gather of records" if world > 1 else ""),
                   "streams_per_gpu": n, "order": args.order, "out_bytes_per_gpu": out_bytes, "in_bytes_per_gpu": in_bytes,
                   "lanes_per_stream": 1 if "lean" in kernel else 65 if "wave" in kernel else args.lanes,
                   "parallelism": "%d independent shards" % world},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "algorithmic_bytes_per_launch": algo, "kernel_ms": k_ms, "check_kernel_ms": c_ms,
                     "measured_copy_GBs": copy_gbs,
                     "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None,
                     "ceilings": ceilings},
    }
    series = os.path.join(ROOT, "profiles", "r03", "stream_series.json")
    if not os.path.exists(series):
        series = os.path.join(ROOT, "profiles", "r02", "stream_series.json")
    if os.path.exists(series):
        # single-GPU kernel time over batch sizes: what one shard of a strong-scaling run costs
        fl = json.load(open(series))
        line["strong_scaling_floor"] = {"what": fl["what"], "source": os.path.relpath(series, ROOT),
                                        "kernel_ms_by_streams": {str(r["streams"]): r["kernel_ms"] for r in fl["rows"]}}
    if rank == 0 and world == 1 and args.cpu_sample:
        # threads: this job's CPU share (OMP_NUM_THREADS is set to it on the GPU box, where os.cpu_count()
        # counts the whole machine); host_cpus says what the machine has
        share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or share
        sample = min(args.cpu_sample, max(n, 256))
        line["cpu_baseline"] = cpu_baseline(distinct, order, cores, sample)
        line["cpu_baseline"]["host_cpus"] = os.cpu_count()
        line["cpu_baseline"]["cpu_share"] = share
        line["cpu_baseline_1thread"] = cpu_baseline(distinct, order, 1, max(256, sample // 16))
    if strong and world > 1 and not args.no_weak:
        # the weak-scaling figure beside it (every rank its own --streams batch), from a second timed loop
        del batch
        m["batch"] = None
        torch.cuda.empty_cache()
        mw = prepare(False)
        dtw = timed(mw)
        line["weak_scaling"] = {"value": float(mw["want"]["got"].sum()) * world / (dtw / args.steps) / 1e9, "unit": "GB/s",
                                "ms_per_step": dtw / args.steps * 1e3, "streams_per_gpu": mw["n"],
                                "what": "every rank decodes its own batch of --streams streams; same step otherwise"}
        line["verified"] = verified[0]
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

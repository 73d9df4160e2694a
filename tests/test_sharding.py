"""CPU tier: host logic of the multi-GPU path (world_size-2 gloo)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from brotli_amd.sharding import all_gather_records, balanced_shards


def test_balanced_shards_cover_and_balance():
    rng = np.random.RandomState(1)
    w = rng.randint(15000, 30000, size=1000)
    for k in (1, 2, 3, 4, 8):
        sh = balanced_shards(w, k)
        assert sh[0][0] == 0 and sh[-1][1] == len(w)
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        sums = [w[a:b].sum() for a, b in sh]
        assert max(sums) - min(sums) <= 2 * w.max()
    assert balanced_shards([5, 5], 4)[-1][1] == 2
    assert balanced_shards([], 2) == [(0, 0), (0, 0)]


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = counts[rank]
    local = torch.full((n * 32,), rank + 1, dtype=torch.uint8)
    out = all_gather_records(local, dist)
    known = all_gather_records(local, dist, sizes=[c * 32 for c in counts])  # shard sizes known: no count exchange
    assert torch.equal(out, known)
    with pytest.raises(ValueError):
        all_gather_records(local, dist, sizes=[c * 32 + 32 for c in counts])
    q.put((rank, out.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(5, 5), (3, 7)])
def test_all_gather_records_gloo_world2(counts):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    want = bytes([1]) * (counts[0] * 32) + bytes([2]) * (counts[1] * 32)
    assert got[0] == want and got[1] == want


def test_bench_strong_scaling_shards_cover_the_batch_once():
    # bench.py --scaling strong: every rank builds the same 65,536-stream order and takes its contiguous shard,
    # cut by compressed bytes; the shards tile the batch, the record sizes handed to the all-gather match them
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    meta, distinct = bench.load_distinct()
    full = bench.build_order(65536, len(distinct), 0x5EED, "shuffled")
    assert sorted(np.bincount(full, minlength=len(distinct))) == [256] * 256
    weights = [len(distinct[j]) for j in full]
    for world in (1, 2, 4, 8):
        shards = balanced_shards(weights, world)
        assert shards[0][0] == 0 and shards[-1][1] == 65536
        assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
        sums = [sum(weights[a:b]) for a, b in shards]
        assert max(sums) - min(sums) <= 2 * max(weights)


def test_all_gather_records_group_of_one_runs_the_collective():
    """No shortcut at world size 1: the same calls as on N ranks (gloo here, nccl = RCCL in the GPU tier)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, port, (6,), q))
    p.start()
    rank, got = q.get(timeout=120)
    p.join(60)
    assert got == bytes([1]) * (6 * 32)

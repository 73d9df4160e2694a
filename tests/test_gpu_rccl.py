"""GPU tier: the record all-gather of the N > 1 path through RCCL (torch backend "nccl") on the one GPU of the
box: a group of one rank, device-resident records from a real decode, the same call bench.py makes per step."""
import os
import socket
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

_RANK0 = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import brotli_amd as ba, helpers as H
from brotli_amd.device import DeviceBatch, RECORD_DTYPE
from brotli_amd.sharding import all_gather_records, balanced_shards
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
meta, distinct = H.packed_streams("cfg2_streams")
streams = [distinct[(11 * j + 5) %% 256] for j in range(512)]
(lo, hi), = balanced_shards([len(s) for s in streams], 1)
assert (lo, hi) == (0, 512)
b = DeviceBatch(streams, [65536] * len(streams))
b.decode(check=ba.CHECK_XXH64)
for sizes in (None, [b.rec.numel()]):
    out = all_gather_records(b.rec, dist, sizes=sizes)          # all_gather_into_tensor on device memory: RCCL
    assert out.is_cuda and out.data_ptr() != b.rec.data_ptr()
    torch.cuda.synchronize()
    r = out.cpu().numpy().view(RECORD_DTYPE)
    for i in range(len(streams)):
        m = meta["streams"][(11 * i + 5) %% 256]
        assert (int(r["rc"][i]), int(r["got"][i]), int(r["used"][i]), int(r["check"][i])) == (0, m["got"], m["len"], int(m["xxh64"], 16)), i
# ragged shards take the padded path: one rank, same code
out = all_gather_records(b.rec[:96], dist)
assert torch.equal(out, b.rec[:96])
t = torch.ones(1, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # bench.py's max-over-ranks of the step time
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""


def test_record_all_gather_over_rccl_group_of_one():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _RANK0 % (H.ROOT, os.path.join(H.ROOT, "tests"))],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

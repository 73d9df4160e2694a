"""Device-resident batches: torch tensors are used only as owners of HBM and
for the stream handle; all work is done by libyeast_mi355x.so through the
C-ABI of include/yeast_batch.h (yeast_batch_device / yeast_check_device).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import CHECK_NONE, SRC_PAD, Record, _err, lib

RECORD_DTYPE = np.dtype([("rc", "<i4"), ("flags", "<u4"), ("got", "<u8"), ("used", "<u8"), ("check", "<u8")])
assert RECORD_DTYPE.itemsize == 32


def pack_streams(streams: Sequence[bytes]):
    """Host-side packing: one contiguous uint8 array (+SRC_PAD zero bytes) and n+1 offsets."""
    lens = np.fromiter((len(s) for s in streams), dtype=np.uint64, count=len(streams))
    off = np.zeros(len(streams) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    buf = np.zeros(int(off[-1]) + SRC_PAD, dtype=np.uint8)
    for s, o in zip(streams, off[:-1]):
        buf[int(o):int(o) + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return buf, off


def plan_output(capacities: Sequence[int], align: int = 64):
    """n+1 output offsets: stream i owns [off[i], off[i+1]); regions are align-byte aligned."""
    cap = (np.asarray(capacities, dtype=np.uint64) + np.uint64(align - 1)) // np.uint64(align) * np.uint64(align)
    off = np.zeros(len(cap) + 1, dtype=np.uint64)
    np.cumsum(cap, out=off[1:])
    return off


class DeviceBatch:
    """A batch resident in HBM: packed input, offsets, output arena, records."""

    def __init__(self, streams: Sequence[bytes], capacities: Sequence[int], device="cuda", align: int = 64):
        """align: alignment of the output regions (1: back to back, every region exactly its capacity)"""
        import torch
        self.torch = torch
        self.n = len(streams)
        buf, soff = pack_streams(streams)
        doff = plan_output(capacities, align)
        self.src = torch.from_numpy(buf).to(device)
        self.src_off = torch.from_numpy(soff.view(np.int64)).to(device)
        self.dst_off = torch.from_numpy(doff.view(np.int64)).to(device)
        self.dst = torch.empty(int(doff[-1]) + 64, dtype=torch.uint8, device=device)
        self.rec = torch.zeros(self.n * 32, dtype=torch.uint8, device=device)
        self.host_src_off = soff
        self.host_dst_off = doff

    def decode(self, check: int = CHECK_NONE, lanes: int = 0):
        """Enqueue decode (+check) on torch's current stream."""
        st = self.torch.cuda.current_stream().cuda_stream
        e = lib().yeast_batch_device(self.src.data_ptr(), self.src_off.data_ptr(), self.dst.data_ptr(),
                                     self.dst_off.data_ptr(), self.rec.data_ptr(), self.n, check, lanes, st)
        if e:
            raise _err("yeast_batch_device")

    def decode_static(self, check: int = CHECK_NONE):
        """The same through yeast_batch_device_static(): the entry point that allocates nothing."""
        st = self.torch.cuda.current_stream().cuda_stream
        e = lib().yeast_batch_device_static(self.src.data_ptr(), self.src_off.data_ptr(), self.dst.data_ptr(),
                                            self.dst_off.data_ptr(), self.rec.data_ptr(), self.n, check, st)
        if e:
            raise _err("yeast_batch_device_static")

    def check(self, kind: int):
        st = self.torch.cuda.current_stream().cuda_stream
        e = lib().yeast_check_device(self.dst.data_ptr(), self.dst_off.data_ptr(), self.rec.data_ptr(), self.n,
                                     kind, st)
        if e:
            raise _err("yeast_check_device")

    def records(self) -> np.ndarray:
        return self.rec.cpu().numpy().view(RECORD_DTYPE)

    def output(self, i: int) -> bytes:
        r = self.records()[i]
        o = int(self.host_dst_off[i])
        return bytes(self.dst[o:o + int(r["got"])].cpu().numpy())

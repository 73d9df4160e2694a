"""The entry point that allocates nothing (include/yeast_batch.h: yeast_batch_device_static) against the
entry point it is a variant of, through the C-ABI: same records, same bytes, on two calls in a row, the second
with a smaller input.  Collected last: a failure here must not hide the other GPU tests."""
import numpy as np
import pytest

import brotli_amd as ba
import helpers as H

pytestmark = pytest.mark.gpu


def test_static_entry_point_matches_the_allocating_one():
    import torch
    from brotli_amd.device import DeviceBatch
    assert ba.lib().yeast_init() == 0
    meta, distinct = H.packed_streams("cfg2_streams")
    _, misc = H.packed_streams("misc_streams")
    crafted = [bytes.fromhex(v["hex"]) for v in H.load_json("crafted.json").values()]
    first = [distinct[(5 * j + 1) % 256] for j in range(700)] + misc + crafted
    second = [distinct[(3 * j + 2) % 256] for j in range(90)] + crafted[:7]
    for streams in (first, second):
        caps = [max(H.oracle_yeast(s)[1], 1) + 64 for s in streams]
        want = DeviceBatch(streams, caps)
        want.dst.fill_(0xAA)
        want.decode(check=ba.CHECK_XXH64, lanes=64)
        have = DeviceBatch(streams, caps)
        have.dst.fill_(0xAA)
        before = torch.cuda.memory_allocated()
        have.decode_static(check=ba.CHECK_XXH64)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        a, b = want.records(), have.records()
        assert a.tobytes() == b.tobytes()
        off = have.host_dst_off
        x, y = want.dst.cpu().numpy(), have.dst.cpu().numpy()
        for i in range(len(streams)):
            g = int(a["got"][i])
            assert np.array_equal(x[int(off[i]):int(off[i]) + g], y[int(off[i]):int(off[i]) + g]), i
    # more streams than the static workspace is laid out for: refused, with a message
    big = DeviceBatch([distinct[0]] * 4, [65536] * 4)
    e = ba.lib().yeast_batch_device_static(big.src.data_ptr(), big.src_off.data_ptr(), big.dst.data_ptr(),
                                           big.dst_off.data_ptr(), big.rec.data_ptr(), 16385, 0, None)
    assert e == -1 and b"at most" in ba.lib().yeast_last_error()

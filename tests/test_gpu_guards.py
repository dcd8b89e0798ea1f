"""Guards named after two code-generation faults of the compiler that the kernels
work around: if a toolchain change brings either back, the test that fails
says which.

1. A lane shift (DPP row_shr:1, zero fill) folded into the subtraction that
   consumes it (v_subrev_u32 with a DPP operand) gave wrong differences.  The
   one place that subtracts a shifted value is the selection of four blocks per
   wavefront, where a lane works out the gap between its first exception and
   the last exception of the lanes before it (integer_codec.h:338-351); the
   shift is kept an instruction of its own there.
2. A wave-uniform 64-bit `<` was compiled into a vector compare paired with a
   scalar select on a stale condition bit.  The number of values in a block,
   min(n - 256 * block, 256), is such a comparison; it is done on 32-bit halves.
"""
import numpy as np
import pytest

import datagen as dg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    import oroch_amd

    rc = oroch_amd.lib.oroch_hip_init(0)
    assert rc == 0, oroch_amd.lib.oroch_hip_strerror(rc)
    return oroch_amd


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("dt", [np.int32, np.uint32, np.int16, np.uint8])
def test_guard_lane_shift_feeding_a_subtraction(hip, oracle, dt):
    """Blocks whose index lists hinge on the gap that reaches across lanes: two or
    three exceptions placed so that the gap in front of a lane's first exception
    (the shifted value minus the lane's own position) decides the width of the
    index list, the length of its varints and with them the block's size."""
    rng = np.random.default_rng(5)
    W = np.dtype(dt).itemsize * 8
    body, big = (1 << (W * 5 // 8)) - 1, 1 << (W - 2)
    blocks = []
    # every pair of lanes of a 16-lane group (values 16 l .. 16 l + 15 belong to lane l),
    # first and last value of the lanes, and a third exception in between for some
    for la in range(16):
        for lb in range(la, 16):
            for pa, pb in ((0, 15), (15, 0), (7, 8)):
                a = rng.integers(0, body + 1, 256).astype(np.int64)
                a[16 * la + pa] += big
                a[16 * lb + pb] += big
                if (la + lb) % 3 == 0 and lb - la > 1:
                    a[16 * ((la + lb) // 2) + 3] += big
                blocks.append(a.astype(dt))
    values = np.concatenate(blocks)
    desc = hip.group_select(dev(values))
    stream, offs = hip.group_encode(dev(values))
    want, want_offs = oracle.blocks_encode(values, 256, 16)
    assert np.array_equal(host(offs).astype(np.uint64), want_offs)
    assert np.array_equal(host(stream), want)
    patched = 0
    for b, blk in enumerate(blocks):
        m = oracle.select(blk)
        assert int(desc[b]["encoding"]) == m.encoding and int(desc[b]["dataspace"]) == m.dataspace, b
        if m.encoding == 6:
            patched += 1
            assert int(desc[b]["noutliers"]) == m.noutliers, b
            assert int(desc[b]["index_encoding"]) == m.index_encoding, b
            if m.index_encoding == 4:
                assert int(desc[b]["index_nbits"]) == m.index_nbits, b
    assert patched > len(blocks) // 2


def test_guard_uniform_64_bit_compare_in_block_count(hip, oracle):
    """An array of more than 2^32 values with a short last block: for most blocks
    the values that remain exceed 32 bits (the high half decides), for the last
    few the low half does, for the last one it is below a block."""
    import torch

    n = (1 << 32) + 256 * 3 + 77
    x = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for lo in range(0, n, step):
        m = min(step, n - lo)
        i = torch.arange(lo, lo + m, dtype=torch.int64, device="cuda")
        # seven bits of a cheap hash, a few constant blocks and a ramp near the end
        v = ((i * 2654435761) >> 13) & 0x7f
        v = torch.where((i >> 8) % 5 == 0, (i >> 8) & 0xff, v)
        x[lo:lo + m] = v.to(torch.uint8)
    nb = (n + 255) // 256
    stream, offs = hip.group_encode(x)
    assert offs.numel() == nb + 1
    o = host(offs).astype(np.uint64)
    assert int(o[-1]) == stream.numel() and np.all(np.diff(o.astype(np.int64)) > 0)
    # first blocks, blocks either side of the 2^32-value mark, and the tail, against the oracle
    for first, count in ((0, 8), ((1 << 24) - 4, 8), (nb - 6, 6)):
        a = host(x[first * 256:min(n, (first + count) * 256)])
        want, want_offs = oracle.blocks_encode(a, 256, 16)
        lo, hi = int(o[first]), int(o[first + count])
        assert hi - lo == want.size, (first, hi - lo, want.size)
        assert np.array_equal(host(stream[lo:hi]), want), first
        assert np.array_equal(o[first:first + count + 1] - o[first], want_offs), first
    desc = hip.group_select(x[(nb - 2) * 256:])
    assert int(desc[-1]["nvalues"]) == 77 and int(desc[0]["nvalues"]) == 256
    back = hip.group_decode(stream, offs, n, np.uint8)
    assert torch.equal(back, x)

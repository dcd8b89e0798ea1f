"""Host buffer in, host buffer out through the overlapped pipeline (oroch_pipe_*):
the same stream and offsets as one pass over the whole array, for chunks small
enough that the tests cross many chunk boundaries."""
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


@pytest.mark.parametrize("dt", [np.int8, np.uint16, np.int32, np.uint32, np.int64, np.uint64])
def test_chunked_pipeline_matches_one_pass(hip, oracle, dt):
    rng = np.random.default_rng(7)
    a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(97)] + [dg.fuzz_block(rng, dt, 131)])
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    # 6 KiB of values per chunk: 3 blocks of int64 ... 24 blocks of int8
    pipe = hip.Pipe(6144)
    for n in (a.size, 256 * 24, 256 * 24 + 1, 255, 1, 0):
        stream, offs = pipe.encode(a[:n])
        e, eo = oracle.blocks_encode(a[:n], 256, 16)
        assert np.array_equal(offs, eo), n
        assert np.array_equal(stream, e), n
        if n:
            back = pipe.decode(stream, offs, n, dt)
            assert np.array_equal(back, a[:n]), n
    # too little room: the length is still reported
    small = np.empty(exp.size - 16, np.uint8)
    with pytest.raises(hip.OrochError):
        pipe.encode(a, out=small)
    # a table that goes backwards is refused
    bad = exp_offs.copy()
    bad[5], bad[6] = bad[6], bad[5]
    with pytest.raises(hip.OrochError):
        pipe.decode(exp, bad, a.size, dt)
    pipe.close()


def test_pipeline_with_pinned_buffers_and_default_pipe(hip, oracle):
    import torch

    n = (1 << 22) + 1000
    a = dg.outlier_mix(n, np.int32, seed=21, shift=-200)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    pipe = hip.Pipe(1 << 20)  # 17 chunks
    pa = torch.from_numpy(a).pin_memory()
    pout = torch.empty(hip.group_bound(np.int32, n), dtype=torch.uint8).pin_memory()
    stream, offs = pipe.encode(pa.numpy(), out=pout.numpy())
    assert np.array_equal(offs, exp_offs) and np.array_equal(stream, exp)
    pback = torch.empty(n, dtype=torch.int32).pin_memory()
    back = pipe.decode(stream, offs, n, np.int32, out=pback.numpy())
    assert np.array_equal(back, a)
    # pageable in, pinned out and the other way round
    stream2, offs2 = pipe.encode(a, out=pout.numpy())
    assert np.array_equal(stream2, exp) and np.array_equal(offs2, exp_offs)
    assert np.array_equal(pipe.decode(exp, exp_offs, n, np.int32, out=pback.numpy()), a)
    pipe.close()
    # the *_host entry points run through the library's own pipe (64 MiB chunks)
    s3, o3 = hip.group_encode_host(a)
    assert np.array_equal(s3, exp) and np.array_equal(o3, exp_offs)
    assert np.array_equal(hip.group_decode_host(s3, o3, n, np.int32), a)

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
    # the *_host entry points run through the library's own pipe
    s3, o3 = hip.group_encode_host(a)
    assert np.array_equal(s3, exp) and np.array_equal(o3, exp_offs)
    assert np.array_equal(hip.group_decode_host(s3, o3, n, np.int32), a)


def test_library_pipe_is_sized_for_the_request_and_can_be_released(hip, oracle):
    """The pipe behind the *_host calls starts small (a block of values must not
    cost hundreds of MiB), is replaced by a larger one when a larger array comes,
    and goes away on request; results are the same throughout."""
    import torch

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    hip.lib.oroch_hip_release_host_buffers()
    before = free_bytes()
    small = dg.outlier_mix(200, np.int16, seed=3, shift=-9)
    e, eo = oracle.blocks_encode(small, 256, 16)
    s, o = hip.group_encode_host(small)
    assert np.array_equal(s, e) and np.array_equal(o, eo)
    assert np.array_equal(hip.group_decode_host(s, o, small.size, np.int16), small)
    held_small = before - free_bytes()
    assert held_small < (64 << 20), held_small
    n = (1 << 23) + 77   # 32 MiB of values: a pipe with larger chunks takes over
    big = dg.outlier_mix(n, np.int32, seed=4, shift=-200)
    e, eo = oracle.blocks_encode(big, 256, 16)
    s, o = hip.group_encode_host(big)
    assert np.array_equal(s, e) and np.array_equal(o, eo)
    assert np.array_equal(hip.group_decode_host(s, o, n, np.int32), big)
    held_big = before - free_bytes()
    assert held_big > held_small
    # small again: served by the pipe that is there
    s, o = hip.group_encode_host(small)
    assert np.array_equal(hip.group_decode_host(s, o, small.size, np.int16), small)
    hip.lib.oroch_hip_release_host_buffers()
    assert before - free_bytes() < (8 << 20)
    s, o = hip.group_encode_host(small)
    assert np.array_equal(hip.group_decode_host(s, o, small.size, np.int16), small)
    hip.lib.oroch_hip_release_host_buffers()

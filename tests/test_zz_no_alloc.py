"""The batch encoder's entry point that allocates nothing (oroch_group_encode_static):
same stream and offsets as oroch_group_encode, on two calls in a row that share
the library's fixed scratch array, the second with a smaller input."""
import ctypes as C

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


def encode_static(hip, x):
    import torch

    code = hip.dtype_code(x.dtype)
    n = x.numel()
    nb = (n + 255) // 256
    bound = hip.lib.oroch_group_bound(code, n)
    out = torch.full((max(bound, 16),), 0xA5, dtype=torch.uint8, device=x.device)
    offs = torch.full((nb + 1,), -1, dtype=torch.int64, device=x.device)
    rc = hip.lib.oroch_group_encode_static(code, x.data_ptr(), n, out.data_ptr(), bound, offs.data_ptr(),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, hip.lib.oroch_hip_strerror(rc)
    torch.cuda.synchronize()
    total = int(offs[nb].item())
    return out[:total].cpu().numpy(), offs.cpu().numpy()


@pytest.mark.parametrize("dt", [np.int32, np.uint16, np.int64])
def test_static_scratch_entry_point_matches_the_allocating_one(hip, dt):
    import torch

    code = hip.dtype_code(dt)
    assert hip.lib.oroch_group_encode_static_max(code) == 1 << 24
    rng = np.random.default_rng(20 + code)
    bits = dict(body_bits=10, out_bits=28) if np.dtype(dt).itemsize >= 4 else dict(body_bits=6, out_bits=15)
    first = np.concatenate([dg.outlier_mix(256 * 3000 + 77, dt, seed=21, shift=-200 if np.dtype(dt).kind == "i" else 0, **bits)] +
                           [dg.fuzz_block(rng, dt, 256) for _ in range(64)])
    second = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(37)] + [dg.fuzz_block(rng, dt, 19)])
    for a in (first, second):  # the second call is smaller and reuses the scratch array as the first left it
        x = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        want_stream, want_offs = hip.group_encode(x)
        got_stream, got_offs = encode_static(hip, x)
        assert np.array_equal(got_offs, want_offs.cpu().numpy())
        assert np.array_equal(got_stream, want_stream.cpu().numpy())


def test_static_scratch_entry_point_limits(hip):
    import torch

    x = torch.zeros(16, dtype=torch.int32, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    offs = torch.full((2,), 7, dtype=torch.int64, device="cuda")
    # beyond the fixed maximum: refused before anything is touched
    rc = hip.lib.oroch_group_encode_static(4, x.data_ptr(), (1 << 24) + 1, out.data_ptr(), 64, offs.data_ptr(), None)
    assert rc == -8 and b"scratch" in hip.lib.oroch_hip_strerror(rc)
    assert hip.lib.oroch_group_encode_static(99, x.data_ptr(), 16, out.data_ptr(), 64, offs.data_ptr(), None) == -1
    # nothing to encode
    assert hip.lib.oroch_group_encode_static(4, None, 0, out.data_ptr(), 64, offs.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(offs[0].item()) == 0

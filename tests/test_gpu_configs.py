"""The benchmark configurations at their full sizes, checked through properties
that do not need a CPU pass over the whole input (round trips, exception lists
rebuilt with tensor ops, shard streams equal to slices of the whole stream),
plus byte comparison with the oracle on a prefix of each input."""
import os

import numpy as np
import pytest

import datagen as dg
import devgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    import oroch_amd

    rc = oroch_amd.lib.oroch_hip_init(0)
    assert rc == 0, oroch_amd.lib.oroch_hip_strerror(rc)
    return oroch_amd


def host(t):
    return t.cpu().numpy()


def test_uint32_1m_bitpck_13_bits(hip, oracle):
    import torch

    n = 1 << 20
    a = dg.uniform_bits(n, np.uint32, 1, 13)
    xs = torch.from_numpy(a.view(np.int32)).cuda()
    assert torch.equal(xs, devgen.uniform_bits(n, torch.int32, 1, 13, xs.device))
    b = hip.bitpck_encode(xs.view(torch.uint32), 13)
    assert b.numel() == 1864144  # 16 * ceil(2^20 / 9)
    assert np.array_equal(host(b), oracle.bitpck_encode(a, 13))
    assert torch.equal(hip.bitpck_decode(b, n, 13, np.uint32).view(torch.int32), xs)


def test_int32_64m_zigzag_and_frame_of_reference(hip, oracle):
    import torch

    n = 1 << 26
    dev = torch.device("cuda", 0)
    for shift in (-20000 + 1000000, -20000):
        x = devgen.uniform_bits(n, torch.int32, 2, 16, dev, shift=shift)
        lo, hi = int(x.min()), int(x.max())
        zz = max((lo << 1) ^ (lo >> 31), (hi << 1) ^ (hi >> 31)) & 0xffffffff
        nb_pck, nb_for = zz.bit_length(), (hi - lo).bit_length()
        assert nb_for == 16
        head = 1 << 20
        a = host(x[:head])
        assert np.array_equal(a, dg.uniform_bits(head, np.int32, 2, 16, shift=shift))
        # a multiple of every capacity, so the packed prefix is the packing of the prefix
        cut = 27720 * 32
        p = hip.bitpck_encode(x, nb_pck)
        assert p.numel() == hip.lib.oroch_bitpck_space(n, nb_pck)
        want = oracle.bitpck_encode(a[:cut], nb_pck)
        assert np.array_equal(host(p[:want.size]), want)
        assert torch.equal(hip.bitpck_decode(p, n, nb_pck, np.int32), x)
        f = hip.bitfor_encode(x, lo, nb_for)
        assert f.numel() == 16 * ((n + 7) // 8)
        want = oracle.bitfor_encode(a[:cut], lo, nb_for)
        assert np.array_equal(host(f[:want.size]), want)
        assert torch.equal(hip.bitfor_decode(f, n, lo, nb_for, np.int32), x)


def test_uint32_256m_one_percent_outliers(hip, oracle):
    import torch

    n = 1 << 28
    dev = torch.device("cuda", 0)
    x64 = devgen.outlier_mix(n, torch.int64, 3, dev)
    xs = x64.to(torch.int32)
    xu = xs.view(torch.uint32)
    head = 1 << 18
    a = host(xs[:head]).view(np.uint32)
    assert np.array_equal(a, dg.outlier_mix(head, np.uint32, 3))
    # whole-array patched packing, origin 0, 10 bits
    b, idx, val, nexc = hip.bitpfr_encode(xu, 0, 10, capacity=n // 64)
    assert nexc == idx.numel()
    where = torch.nonzero(x64 >> 10).flatten()
    assert idx.numel() == where.numel() and 0.009 * n < idx.numel() < 0.011 * n
    pos = torch.cumsum(idx.to(torch.int64) + 1, 0) - 1
    assert torch.equal(pos, where)
    assert torch.equal(val.to(torch.int64), x64[where] >> 10)
    cut = 12 * 4096
    wb, wi, wv = oracle.bitpfr_encode(a[:cut], 0, 10)
    assert np.array_equal(host(b[:wb.size]), wb)
    assert np.array_equal(host(idx[:wi.size]).astype(np.uint64), wi)
    back = hip.bitpfr_decode(b, n, 0, 10, idx, val, np.uint32)
    assert torch.equal(back.view(torch.int32), xs)
    del b, idx, val, back, where, pos
    # the same input through per-block selection
    stream, offs = hip.group_encode(xu)
    want, want_offs = oracle.blocks_encode(a, 256, 16)
    assert np.array_equal(host(offs[:want_offs.size]).astype(np.uint64), want_offs)
    assert np.array_equal(host(stream[:want.size]), want)
    back = hip.group_decode(stream, offs, n, np.uint32)
    assert torch.equal(back.view(torch.int32), xs)


def test_int64_1g_block_mix_and_shards(hip, oracle):
    import torch
    from oroch_amd import sharding

    n = 1 << 30
    dev = torch.device("cuda", 0)
    x = devgen.block_mix_int64(n, 4, dev)
    stream, offs = hip.group_encode(x)
    head = 1 << 16
    a = host(x[:head])
    want, want_offs = oracle.blocks_encode(a, 256, 16)
    assert np.array_equal(host(offs[:want_offs.size]).astype(np.uint64), want_offs)
    assert np.array_equal(host(stream[:want.size]), want)
    encs = set(int(e) for e in hip.group_select(x[:1 << 20])["encoding"])
    assert {0, 1, 6} <= encs and encs & {4, 5}, encs
    back = hip.group_decode(stream, offs, n, np.int64)
    assert torch.equal(back, x)
    del back
    # a shard encoded on its own is a slice of the whole stream
    for world, rank in ((2, 1), (4, 2), (8, 5)):
        lo, hi = sharding.shard_range(n, rank, world)
        s, o = hip.group_encode(x[lo:hi])
        b0, b1 = int(offs[lo // 256]), int(offs[(hi + 255) // 256])
        assert s.numel() == b1 - b0
        assert torch.equal(s, stream[b0:b1])
        assert torch.equal(o, offs[lo // 256:(hi + 255) // 256 + 1] - b0)
        del s, o


def test_int16_and_uint64_host_to_device_to_host(hip, oracle):
    import torch

    dev = torch.device("cuda", 0)
    # int16: 9-bit signed noise with 0.5 % full-range values
    n = 1 << 28
    a = host(devgen.outlier_mix(n, torch.int16, 5, dev, body_bits=9, out_bits=16, permille=5,
                                shift=-256))
    stream, offs = hip.group_encode_host(a)
    want, want_offs = oracle.blocks_encode(a[:1 << 18], 256, 16)
    assert np.array_equal(offs[:want_offs.size], want_offs)
    assert np.array_equal(stream[:want.size], want)
    back = hip.group_decode_host(stream, offs, n, np.int16)
    assert np.array_equal(back, a)
    del a, back, stream
    # uint64: 20-bit values above 2^40 with 1 % 50-bit outliers
    n = 1 << 27
    a = host(devgen.outlier_mix(n, torch.int64, 6, dev, body_bits=20, out_bits=50,
                                shift=1 << 40)).view(np.uint64)
    stream, offs = hip.group_encode_host(a)
    want, want_offs = oracle.blocks_encode(a[:1 << 18], 256, 16)
    assert np.array_equal(offs[:want_offs.size], want_offs)
    assert np.array_equal(stream[:want.size], want)
    back = hip.group_decode_host(stream, offs, n, np.uint64)
    assert np.array_equal(back, a)


def _pfr_shard_worker(rank, world, port, n, nbits, outdir):
    import torch
    import torch.distributed as dist

    import oroch_amd
    from oroch_amd import sharding

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)  # the ranks share the one GPU of this box
        assert oroch_amd.lib.oroch_hip_init(0) == 0
        a = dg.outlier_mix(n, np.uint32, seed=7, body_bits=nbits, permille=3)
        lo, hi = sharding.whole_array_range(n, rank, world, nbits)
        x = torch.from_numpy(a[lo:hi].view(np.int32)).cuda().view(torch.uint32)
        packed, gaps, highs, base = sharding.bitpfr_encode_shard(x, lo, 0, nbits)
        np.savez(os.path.join(outdir, "gpfr%d.npz" % rank), packed=packed.cpu().numpy(),
                 gaps=gaps.cpu().numpy().astype(np.uint64), highs=highs.cpu().numpy().astype(np.uint64),
                 base=np.uint64(base))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_whole_array_bitpfr_sharded_on_the_gpu(hip, oracle, tmp_path, world):
    """Config 3 on more than one GPU: every rank packs its word-aligned range with
    the HIP kernels, two numbers per rank are exchanged, and the shards end to end
    are the oracle's code of the whole array (ranks share this box's one GPU)."""
    import socket

    import torch.multiprocessing as mp

    n, nbits = 300007, 10
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_pfr_shard_worker, args=(world, port, n, nbits, str(tmp_path)), nprocs=world, join=True)
    a = dg.outlier_mix(n, np.uint32, seed=7, body_bits=nbits, permille=3)
    want_packed, want_gaps, want_highs = oracle.bitpfr_encode(a, 0, nbits)
    parts = [np.load(os.path.join(str(tmp_path), "gpfr%d.npz" % r)) for r in range(world)]
    assert [int(p["base"]) for p in parts] == list(np.cumsum([0] + [p["gaps"].size for p in parts[:-1]]))
    assert np.array_equal(np.concatenate([p["packed"] for p in parts]), want_packed)
    assert np.array_equal(np.concatenate([p["gaps"] for p in parts]), want_gaps)
    assert np.array_equal(np.concatenate([p["highs"] for p in parts]), want_highs)

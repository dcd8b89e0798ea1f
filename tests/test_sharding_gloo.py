"""World-size-2 (and 3) CPU rehearsal of the multi-GPU path: each rank owns a
contiguous range of blocks, encodes it on its own (here with the CPU oracle in
place of the GPU), and one byte count per rank is all that is exchanged.  The
concatenation of the shard streams, with shard-local offsets rebased, must be
the stream one process produces for the whole array."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, outdir):
    import datagen as dg
    from pyoracle import oracle
    from oroch_amd import sharding

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        a = dg.outlier_mix(n, np.int32, seed=5, shift=-200)
        lo, hi = sharding.shard_range(n, rank, world)
        assert lo % sharding.BLOCK_VALUES == 0
        stream, offs = oracle.blocks_encode(a[lo:hi], 256, 16)
        counts = sharding.gather_byte_counts(stream.size, torch.device("cpu"))
        assert len(counts) == world and counts[rank] == stream.size
        base = sharding.shard_base(counts, rank)
        t = sharding.max_over_ranks(float(rank + 1), torch.device("cpu"))
        assert t == float(world)
        np.savez(os.path.join(outdir, "shard%d.npz" % rank), stream=stream, offs=offs,
                 base=np.uint64(base), lo=np.uint64(lo), hi=np.uint64(hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 100000), (3, 70001), (2, 300)])
def test_shards_concatenate_to_the_whole_stream(tmp_path, world, n):
    import datagen as dg
    from pyoracle import oracle

    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    a = dg.outlier_mix(n, np.int32, seed=5, shift=-200)
    want, want_offs = oracle.blocks_encode(a, 256, 16)
    parts, offs, covered = [], [], 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "shard%d.npz" % r))
        assert int(z["lo"]) == covered
        covered = int(z["hi"])
        assert int(z["base"]) == sum(len(p) for p in parts)
        parts.append(z["stream"])
        offs.append(z["offs"][:-1] + z["base"])
        last = z["offs"][-1] + z["base"]
    assert covered == n
    got = np.concatenate(parts)
    got_offs = np.concatenate(offs + [np.array([last], np.uint64)])
    assert np.array_equal(got, want)
    assert np.array_equal(got_offs.astype(np.uint64), want_offs)


def test_shard_ranges_tile_the_array():
    from oroch_amd import sharding

    for n in (0, 1, 255, 256, 257, 1 << 20, (1 << 20) + 13):
        for world in (1, 2, 3, 8):
            edges = [sharding.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (lo, hi), (lo2, _) in zip(edges, edges[1:]):
                assert hi == lo2 and lo <= hi and hi % 256 == 0 or hi == n


def _pfr_worker(rank, world, port, n, nbits, outdir):
    import datagen as dg
    from pyoracle import oracle
    from oroch_amd import sharding

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        a = dg.outlier_mix(n, np.uint32, seed=7, body_bits=nbits, permille=3)
        lo, hi = sharding.whole_array_range(n, rank, world, nbits)
        assert lo % (128 // nbits) == 0 or lo == n
        packed, gaps, highs = oracle.bitpfr_encode(a[lo:hi], 0, nbits)
        last = lo + int(np.cumsum(gaps.astype(np.int64) + 1)[-1]) - 1 if gaps.size else -1
        summaries = sharding.gather_exception_summaries(gaps.size, last, torch.device("cpu"))
        assert summaries[rank] == (gaps.size, last)
        if gaps.size:
            gaps = gaps.copy()
            gaps[0] += np.uint64(sharding.first_gap_fixup(summaries, rank, lo))
        np.savez(os.path.join(outdir, "pfr%d.npz" % rank), packed=packed, gaps=gaps, highs=highs,
                 base=np.uint64(sharding.exception_base(summaries, rank)), lo=np.uint64(lo), hi=np.uint64(hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nbits", [(2, 100000, 10), (3, 70001, 13), (2, 5000, 7), (3, 40, 10)])
def test_whole_array_bitpfr_shards_concatenate(tmp_path, world, n, nbits):
    """One whole-sequence bitpfr code (bitpfr.h:95-124) cut over ranks at word
    boundaries: packed words and value lists concatenate, and the gap lists do
    after the first gap of each shard is counted from the last exception before it."""
    import datagen as dg
    from pyoracle import oracle

    port = _free_port()
    mp.spawn(_pfr_worker, args=(world, port, n, nbits, str(tmp_path)), nprocs=world, join=True)
    a = dg.outlier_mix(n, np.uint32, seed=7, body_bits=nbits, permille=3)
    want_packed, want_gaps, want_highs = oracle.bitpfr_encode(a, 0, nbits)
    packed, gaps, highs, covered = [], [], [], 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "pfr%d.npz" % r))
        assert int(z["lo"]) == covered
        covered = int(z["hi"])
        assert int(z["base"]) == sum(g.size for g in gaps)
        packed.append(z["packed"])
        gaps.append(z["gaps"])
        highs.append(z["highs"])
    assert covered == n
    assert np.array_equal(np.concatenate(packed), want_packed)
    assert np.array_equal(np.concatenate(gaps), want_gaps)
    assert np.array_equal(np.concatenate(highs), want_highs)
    back = oracle.bitpfr_decode(np.concatenate(packed), n, 0, nbits, np.concatenate(gaps), np.concatenate(highs),
                                np.uint32)
    assert np.array_equal(back, a)


def test_whole_array_ranges_start_fresh_words():
    from oroch_amd import sharding

    for nbits in (1, 7, 10, 13, 33, 64):
        cap = 128 // nbits
        for n in (0, 1, cap, cap + 1, 100003, 1 << 20):
            for world in (1, 2, 3, 8):
                edges = [sharding.whole_array_range(n, r, world, nbits) for r in range(world)]
                assert edges[0][0] == 0 and edges[-1][1] == n
                for (lo, hi), (lo2, _) in zip(edges, edges[1:]):
                    assert hi == lo2 and lo <= hi and (hi % cap == 0 or hi == n)
    # gaps: nothing before, exceptions two shards back, none in between
    s = [(0, -1), (3, 17), (0, -1), (2, 90)]
    assert sharding.first_gap_fixup(s, 0, 0) == 0
    assert sharding.first_gap_fixup(s, 1, 12) == 12
    assert sharding.first_gap_fixup(s, 2, 24) == 24 - 18
    assert sharding.first_gap_fixup(s, 3, 36) == 36 - 18
    assert [sharding.exception_base(s, r) for r in range(4)] == [0, 0, 3, 3]

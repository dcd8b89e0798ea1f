"""Searching and random access over block images, and the growable compressed
array, against a plain numpy model of the same operations and the CPU oracle's
block images.  Scenarios of the reference's tests/unit/integer_array.cc included.
"""
import numpy as np
import pytest

import datagen as dg

pytestmark = pytest.mark.gpu

ALL = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]


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


def mixed_blocks(rng, dt, nblocks, last):
    """Blocks of every encoding, the last one with `last` values."""
    parts = [dg.fuzz_block(rng, dt, 256) for _ in range(nblocks)]
    if last:
        parts.append(dg.fuzz_block(rng, dt, last))
    return np.concatenate(parts)


def first_positions(a, needles):
    out = np.empty(len(needles), np.uint64)
    for j, v in enumerate(needles):
        hit = np.flatnonzero(a == v)
        out[j] = hit[0] if hit.size else np.uint64((1 << 64) - 1)
    return out


def some_needles(rng, a, dt, count):
    """Values that are there (duplicates included), near misses and values that are not."""
    info = np.iinfo(dt)
    there = a[rng.integers(0, a.size, size=count)]
    near = (there.astype(np.int64 if dt.kind == "i" else np.uint64) + 1).astype(dt)
    edge = np.array([info.min, info.max, 0, 1], dt)
    rnd = dg.fuzz_block(rng, dt, count)
    return np.concatenate([there, near, edge, rnd])


@pytest.mark.parametrize("dt", ALL)
def test_find_and_at_over_block_images(hip, dt):
    dt = np.dtype(dt)
    rng = np.random.default_rng(9100 + dt.itemsize * 2 + (dt.kind == "u"))
    for nblocks, last in ((37, 0), (12, 77), (0, 5), (1, 0)):
        a = mixed_blocks(rng, dt, nblocks, last)
        stream, offs = hip.group_encode(dev(a))
        needles = some_needles(rng, a, dt, 24)
        want = first_positions(a, needles)
        # all at once
        got = hip.group_find(stream, offs, a.size, dt, needles)
        assert np.array_equal(got, want), (dt, nblocks, last)
        # one at a time: blocks that cannot hold the value are skipped
        for v, w in zip(needles[::3], want[::3]):
            assert int(hip.group_find(stream, offs, a.size, dt, [v])[0]) == int(w), (dt, v)
        # the one-block-per-wavefront walker (the only one for 64-bit types) agrees
        with hip.option("find_one_block", 1):
            assert np.array_equal(hip.group_find(stream, offs, a.size, dt, needles), want)
            for v, w in zip(needles[1::3], want[1::3]):
                assert int(hip.group_find(stream, offs, a.size, dt, [v])[0]) == int(w), (dt, v)
        # random access, positions beyond the end included
        pos = np.concatenate([rng.integers(0, a.size, size=200), [0, a.size - 1, a.size, a.size + 999]])
        got = host(hip.group_at(stream, offs, a.size, dt, pos)).view(dt)
        exp = np.where(pos < a.size, a[np.minimum(pos, a.size - 1)], 0).astype(dt)
        assert np.array_equal(got, exp), (dt, nblocks, last)
        # every position once: all encodings, exceptions and their neighbours, both
        # the direct reads and the blocks that have to be decoded
        every = host(hip.group_at(stream, offs, a.size, dt, np.arange(a.size))).view(dt)
        assert np.array_equal(every, a), (dt, nblocks, last, np.flatnonzero(every != a)[:8])


@pytest.fixture
def find_one_block(hip, request):
    with hip.option("find_one_block", request.param):
        yield request.param


@pytest.mark.parametrize("find_one_block", [0, 1], indirect=True)
def test_find_shortcut_cases(hip, find_one_block):
    """Blocks built so that the metadata alone rules the value in or out
    (integer_array.h:85-126): a constant block, narrow zigzag blocks, narrow
    blocks above an origin, and the value just outside each of them."""
    dt = np.dtype(np.int32)
    rng = np.random.default_rng(77)
    const = np.full(256, -123456, dt)
    narrow = rng.integers(-8, 8, size=256).astype(dt)        # bitpck, 4 bits
    lifted = (rng.integers(0, 64, size=256) + 1000000).astype(dt)  # bitfor
    lifted[0], lifted[1] = 1000000, 1000063
    wide = rng.integers(-2**31, 2**31, size=256).astype(dt)   # normal
    a = np.concatenate([narrow, const, lifted, wide, narrow, lifted])
    stream, offs = hip.group_encode(dev(a))
    encs = [int(e) for e in hip.group_select(dev(a))["encoding"]]
    assert encs[1] == 0 and encs[0] == 4 and encs[2] == 5 and encs[3] == 1, encs
    needles = np.array([-123456, -123455, -8, 7, 8, -9, 1000000, 1000063, 1000064, 999999,
                        int(wide[17]), int(narrow[200]), int(lifted[255])], dt)
    want = first_positions(a, needles)
    for v, w in zip(needles, want):
        assert int(hip.group_find(stream, offs, a.size, dt, [v])[0]) == int(w), v
    assert np.array_equal(hip.group_find(stream, offs, a.size, dt, needles), want)


def check_array(hip, oracle, arr, model):
    """The array holds `model`: length, values, block images, tail."""
    model = np.asarray(model)
    assert len(arr) == model.size
    assert np.array_equal(arr.read(), model)
    stream, offs, tail = arr.view()
    nfull = model.size // 256 * 256
    assert np.array_equal(tail, model[nfull:])
    if nfull:
        exp, exp_offs = oracle.blocks_encode(model[:nfull], 256, 16)
        assert np.array_equal(host(offs).astype(np.uint64), exp_offs)
        assert np.array_equal(host(stream), exp)
        # the view is what the stateless entry points take
        back = hip.group_decode(stream, offs, nfull, model.dtype)
        assert np.array_equal(host(back).view(model.dtype), model[:nfull])
    else:
        assert stream is None and offs is None


@pytest.mark.parametrize("dt", ALL)
def test_growable_array(hip, oracle, dt):
    dt = np.dtype(dt)
    rng = np.random.default_rng(5150 + dt.itemsize * 2 + (dt.kind == "u"))
    arr = hip.IntegerArray(dt)
    assert arr.empty() and len(arr) == 0
    assert arr.find(dt.type(0)) == hip.NOT_FOUND
    model = np.empty(0, dt)
    # bulk appends that start and stop anywhere relative to the block boundaries
    for step, count in enumerate((1, 254, 1, 256, 257, 1000, 5000, 3, 0, 767)):
        piece = mixed_blocks(rng, dt, count // 256, count % 256) if count else np.empty(0, dt)
        arr.append(dev(piece) if step % 2 else piece)
        model = np.concatenate([model, piece])
        if step in (0, 2, 4, 9):
            check_array(hip, oracle, arr, model)
    # find / at over groups and tail
    needles = some_needles(rng, model, dt, 16)
    needles = np.concatenate([needles, model[-3:]])
    assert np.array_equal(arr.find(needles), first_positions(model, needles))
    assert arr.find(model[-1]) == int(first_positions(model, model[-1:])[0])
    pos = rng.integers(0, model.size, size=100)
    assert np.array_equal(arr.at(pos), model[pos])
    assert arr.at(model.size - 1) == model[-1] and arr[0] == model[0]
    with pytest.raises(hip.OrochError):
        arr.at(model.size)
    with pytest.raises(hip.OrochError):
        arr.insert(model.size + 1, 0)
    # single insertions: in the tail, at the very end, in a group, at the front
    for where in (model.size - 2, model.size, 4000, 255, 256, 0):
        v = dg.fuzz_block(rng, dt, 1)[0]
        arr.insert(where, int(v))
        model = np.insert(model, where, v)
    check_array(hip, oracle, arr, model)
    # fill the tail up by insertions until it becomes a group
    while len(arr) % 256:
        v = dg.fuzz_block(rng, dt, 1)[0]
        arr.insert(len(arr), int(v))
        model = np.append(model, v)
    check_array(hip, oracle, arr, model)
    assert np.array_equal(arr.read(300, 1000), model[300:1300])
    arr.clear()
    assert arr.empty()
    arr.append(model[:600])
    check_array(hip, oracle, arr, model[:600])


@pytest.mark.parametrize("dt", [np.int8, np.uint16, np.int32, np.uint64])
def test_array_as_container(hip, dt):
    """save() gives the serialised container of the same values, byte for byte;
    load() takes any such container and the array goes on from there."""
    dt = np.dtype(dt)
    rng = np.random.default_rng(404 + dt.itemsize)
    for nblocks, last in ((7, 100), (3, 0), (0, 9), (0, 0)):
        model = mixed_blocks(rng, dt, nblocks, last) if nblocks or last else np.empty(0, dt)
        arr = hip.IntegerArray(dt)
        arr.append(model[:model.size // 2])
        arr.append(model[model.size // 2:])
        blob = arr.save()
        assert np.array_equal(blob, hip.container_encode_host(model)), (dt, nblocks, last)
        other = hip.IntegerArray(dt)
        other.append(mixed_blocks(rng, dt, 2, 5))      # replaced by load
        other.load(blob)
        assert len(other) == model.size and np.array_equal(other.read(), model)
        more = mixed_blocks(rng, dt, 1, 30)
        other.append(more)
        grown = np.concatenate([model, more])
        assert np.array_equal(other.read(), grown)
        if grown.size:
            v = grown[-1]
            assert other.find(v) == int(np.flatnonzero(grown == v)[0])
        assert np.array_equal(other.save(), hip.container_encode_host(grown))
    wrong = hip.IntegerArray(np.int16 if dt != np.int16 else np.int32)
    with pytest.raises(hip.OrochError):
        wrong.load(blob)
    with pytest.raises(hip.OrochError):
        arr.load(blob[:40])


def test_reference_array_scenarios(hip):
    """tests/unit/integer_array.cc: "small integer array" and "large integer array"."""
    arr = hip.IntegerArray(np.int32)
    assert arr.find(np.int32(0)) == hip.NOT_FOUND
    assert arr.find(np.int32(1)) == hip.NOT_FOUND
    arr.insert(0, 0)
    assert arr.at(0) == 0 and arr.find(np.int32(0)) == 0
    arr.insert(0, 1)
    assert arr.at(0) == 1 and arr.at(1) == 0
    assert arr.find(np.int32(0)) == 1 and arr.find(np.int32(1)) == 0

    n = 100000
    arr = hip.IntegerArray(np.int32)
    # (the reference inserts one by one at the end; the first thousand go that way here)
    for i in range(1000):
        arr.insert(i, i)
    arr.append(np.arange(1000, n, dtype=np.int32))
    arr.insert(0, -1)
    assert len(arr) == n + 1
    got = arr.find(np.arange(n, dtype=np.int32))
    assert np.array_equal(got, np.arange(n, dtype=np.uint64) + 1)
    for i in (0, 1, 255, 256, 99999):
        assert arr.find(np.int32(i)) == i + 1
    assert arr.find(np.int32(-1)) == 0


def test_find_and_at_at_scale(hip):
    """2^26 values: a value planted once is found where it is, a value that is
    nowhere is not, random positions read back what was encoded."""
    import torch

    import devgen

    n = 1 << 26
    x = devgen.outlier_mix(n, torch.int32, seed=31, device="cuda", shift=-200)
    x[x == 123456789] = 0
    x[x == -2000000000] = 0
    spot = 50_000_017
    x[spot] = 123456789
    x[spot + 1000] = 123456789
    stream, offs = hip.group_encode(x)
    assert int(hip.group_find(stream, offs, n, np.int32, [123456789])[0]) == spot
    assert int(hip.group_find(stream, offs, n, np.int32, [-2000000000])[0]) == hip.NOT_FOUND
    both = hip.group_find(stream, offs, n, np.int32, [123456789, -2000000000, int(x[0])])
    assert int(both[0]) == spot and int(both[1]) == hip.NOT_FOUND and int(both[2]) == 0
    pos = torch.randint(0, n, (100000,), device="cuda")
    got = hip.group_at(stream, offs, n, np.int32, host(pos))
    assert torch.equal(got, x[pos])


@pytest.mark.parametrize("dt", [np.uint8, np.int16, np.int32, np.uint64])
def test_find_and_at_on_damaged_streams(hip, dt):
    """Searching and random access read files and network buffers too: damaged
    offsets or block bytes may give wrong answers, never a read outside the
    stream (which would show up here as a GPU fault)."""
    import torch

    dt = np.dtype(dt)
    rng = np.random.default_rng(61 + dt.itemsize)
    a = mixed_blocks(rng, dt, 64, 0)
    n = a.size
    stream, offs = hip.group_encode(dev(a))
    total = int(offs[-1])
    needles = some_needles(rng, a, dt, 8)
    pos = np.arange(n)
    o = host(offs).astype(np.uint64)
    bad = o.copy()
    bad[3], bad[4] = bad[4], bad[3]
    bad[10] = np.uint64(total + 4096)
    bad[17] += np.uint64(5)
    bad[30] = np.uint64(2 ** 63 + 12345)
    bad[40] = np.uint64(0)
    bad_offs = torch.from_numpy(bad.astype(np.int64)).cuda()
    got = host(hip.group_at(stream, bad_offs, n, dt, pos)).view(dt)
    ok_blocks = [b for b in range(64) if all(bad[b + d] == o[b + d] for d in (0, 1))]
    for b in ok_blocks[:8] + ok_blocks[-8:]:
        assert np.array_equal(got[b * 256:(b + 1) * 256], a[b * 256:(b + 1) * 256]), b
    hip.group_find(stream, bad_offs, n, dt, needles)
    hip.group_find(stream, bad_offs, n, dt, needles[:1])
    for trial in range(3):
        s = host(stream).copy()
        hits = rng.integers(0, s.size, size=600)
        s[hits] ^= rng.integers(1, 256, size=600).astype(np.uint8)
        ds = torch.from_numpy(s).cuda()
        hip.group_at(ds, offs, n, dt, pos)
        hip.group_find(ds, offs, n, dt, needles)
        hip.group_find(ds, offs, n, dt, needles[:1])
    torch.cuda.synchronize()

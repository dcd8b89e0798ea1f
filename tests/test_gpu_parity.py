"""HIP path (through the C ABI) against the golden fixtures and the CPU oracle.

Byte-exact for every encoded stream, index-exact for exception lists, value
exact for every decode.
"""
import numpy as np
import pytest

import datagen as dg

pytestmark = pytest.mark.gpu

META_TO_DESC = {"encoding": "encoding", "nbits": "nbits", "origin": "origin",
                "dataspace": "dataspace", "metaspace": "metaspace", "noutliers": "noutliers",
                "index_encoding": "index_encoding", "index_nbits": "index_nbits",
                "value_encoding": "value_encoding", "value_nbits": "value_nbits"}


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


def check_desc(desc_row, meta, fields, ctx):
    exp = dict(zip(fields, meta))
    for mf, df in META_TO_DESC.items():
        got = int(desc_row[df])
        want = int(exp[mf])
        if mf in ("index_encoding", "index_nbits", "value_encoding", "value_nbits") and \
                exp["encoding"] != 6:
            continue  # only meaningful for bitpfr
        if mf == "index_nbits" and exp["index_encoding"] != 4:
            # a varint index list carries no width in the stream (integer_codec.h:177-182);
            # the descriptor reports 0 there
            want = 0
        assert got == want, (mf, got, want, ctx)


def test_golden_single_blocks(hip, golden):
    z, cases, fields = golden
    count = 0
    for i, c in enumerate(cases):
        if c["kind"] != "group":
            continue
        a = z["c%d_in" % i].view(dg.DTYPES[c["dtype"]])
        x = dev(a)
        exp = z["c%d_out" % i]
        if a.size == 0:
            stream, offs = hip.group_encode(x)
            assert stream.numel() == 0 and int(offs[0]) == 0
            continue
        d = hip.group_select(x)
        assert d.size == 1
        check_desc(d[0], c["meta"], fields, (i, c))
        stream, offs = hip.group_encode(x)
        got = host(stream)
        assert int(offs[1]) == got.size and got.size == (exp.size + 15) // 16 * 16, (i, c)
        assert np.array_equal(got[:exp.size], exp), (i, c)
        assert not got[exp.size:].any(), (i, c)
        back = host(hip.group_decode(stream, offs, a.size, a.dtype))
        assert np.array_equal(back, a), (i, c)
        count += 1
    assert count > 800


def test_golden_block_streams(hip, golden):
    z, cases, _ = golden
    seen = 0
    for i, c in enumerate(cases):
        if c["kind"] != "blocks":
            continue
        a = z["c%d_in" % i].view(dg.DTYPES[c["dtype"]])
        stream, offs = hip.group_encode(dev(a))
        assert np.array_equal(host(offs).astype(np.uint64), z["c%d_offs" % i]), c
        assert np.array_equal(host(stream), z["c%d_out" % i]), c
        assert np.array_equal(host(hip.group_decode(stream, offs, a.size, a.dtype)), a), c
        seen += 1
    assert seen == 5


def test_golden_whole_sequence(hip, golden):
    z, cases, _ = golden
    seen = set()
    for i, c in enumerate(cases):
        k = c["kind"]
        if k not in ("bitpck", "bitfor", "bitpfr"):
            continue
        seen.add(k)
        dt = np.dtype(dg.DTYPES[c["dtype"]])
        a = z["c%d_in" % i].view(dt)
        exp = z["c%d_out" % i]
        x = dev(a)
        W = dt.itemsize * 8
        if k == "bitpck":
            got = hip.bitpck_encode(x, c["nbits"])
            assert np.array_equal(host(got), exp), c
            back = host(hip.bitpck_decode(got, a.size, c["nbits"], dt))
            if c["nbits"] >= W:
                assert np.array_equal(back, a), c
        elif k == "bitfor":
            got = hip.bitfor_encode(x, c["origin"], c["nbits"])
            assert np.array_equal(host(got), exp), c
        else:
            got, idx, val, nexc = hip.bitpfr_encode(x, c["origin"], c["nbits"])
            assert np.array_equal(host(got), exp), c
            assert nexc == z["c%d_idx" % i].size, c
            assert np.array_equal(host(idx).astype(np.uint64), z["c%d_idx" % i]), c
            assert np.array_equal(host(val).astype(np.uint64), z["c%d_val" % i]), c
            back = host(hip.bitpfr_decode(got, a.size, c["origin"], c["nbits"], idx, val, dt))
            assert np.array_equal(back, a), c
    assert seen == {"bitpck", "bitfor", "bitpfr"}


@pytest.mark.parametrize("dt", dg.DTYPES)
def test_fuzz_blocks_against_oracle(hip, oracle, dt):
    rng = np.random.default_rng(1000 + dg.DTYPES.index(dt))
    nblocks = 3000
    a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(nblocks)] +
                       [dg.fuzz_block(rng, dt, int(rng.integers(1, 256)))])
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    x = dev(a)
    stream, offs = hip.group_encode(x)
    got_offs = host(offs).astype(np.uint64)
    bad = np.nonzero(got_offs != exp_offs)[0]
    if bad.size:
        b = int(bad[0]) - 1
        blk = a[b * 256:(b + 1) * 256]
        d = hip.group_select(dev(blk))[0]
        raise AssertionError("block %d size differs: hip %s oracle %s" %
                             (b, d, oracle.select(blk)))
    got = host(stream)
    if not np.array_equal(got, exp):
        pos = int(np.nonzero(got != exp)[0][0])
        b = int(np.searchsorted(exp_offs, pos, side="right")) - 1
        raise AssertionError("stream differs at byte %d (block %d, +%d): %s" %
                             (pos, b, pos - int(exp_offs[b]), oracle.select(a[b * 256:(b + 1) * 256])))
    # all seven encodings must have been exercised (8-bit types cannot reach varints)
    encs = set(int(e) for e in hip.group_select(x)["encoding"])
    assert encs >= ({0, 1, 4, 5, 6} if np.dtype(dt).itemsize == 1 else set(range(7))), encs
    back = host(hip.group_decode(stream, offs, a.size, dt))
    assert np.array_equal(back, a)


@pytest.mark.parametrize("dt", dg.DTYPES)
def test_one_block_kernels_agree_for_every_type(hip, oracle, dt):
    """The one-block-per-wavefront selection and encoder (the 64-bit types' own
    path; a switch for the others) fetch a wavefront's next block ahead and take
    a short last block after their loop: the same bytes as the oracle for full
    blocks, a short last block, fewer blocks than wavefronts and a single value."""
    rng = np.random.default_rng(2000 + dg.DTYPES.index(dt))
    blocks = [dg.fuzz_block(rng, dt, 256) for _ in range(700)]
    for cut in (700 * 256, 699 * 256 + 31, 5 * 256, 3 * 256 + 255, 256, 255, 1):
        a = np.concatenate(blocks)[:cut]
        exp, exp_offs = oracle.blocks_encode(a, 256, 16)
        x = dev(a)
        with hip.option("select_one_block", 1), hip.option("encode_one_block", 1):
            stream, offs = hip.group_encode(x)
            desc = hip.group_select(x)
        assert np.array_equal(host(offs).astype(np.uint64), exp_offs), cut
        assert np.array_equal(host(stream), exp), cut
        assert int(desc[-1]["nvalues"]) == (cut - 1) % 256 + 1
        assert np.array_equal(host(hip.group_decode(stream, offs, a.size, dt)).view(np.dtype(dt)), a)


@pytest.mark.parametrize("dt", [np.uint32, np.int32, np.int64, np.uint8, np.int16])
def test_whole_sequence_against_oracle(hip, oracle, dt):
    dt = np.dtype(dt)
    W = dt.itemsize * 8
    n = 100003
    for nbits in sorted(b for b in {1, 2, 3, 5, 7, 10, 13, W // 2, W - 1, W} if b <= W):
        a = dg.uniform_bits(n, dt, nbits, W)
        x = dev(a)
        got = hip.bitpck_encode(x, nbits)
        assert np.array_equal(host(got), oracle.bitpck_encode(a, nbits)), nbits
        assert np.array_equal(host(hip.bitpck_decode(got, n, nbits, dt)),
                              oracle.bitpck_decode(host(got), n, nbits, dt)), nbits
        origin = int(a.min())
        got = hip.bitfor_encode(x, origin, nbits)
        assert np.array_equal(host(got), oracle.bitfor_encode(a, origin, nbits)), nbits
        assert np.array_equal(host(hip.bitfor_decode(got, n, origin, nbits, dt)),
                              oracle.bitfor_decode(host(got), n, origin, nbits, dt)), nbits
    for nbits, permille in ((1, 500), (4, 10), (W // 2, 1), (W - 1, 3)):
        a = dg.outlier_mix(n, dt, 9, body_bits=nbits, out_bits=W, permille=permille, shift=-5)
        origin = int(a.min())
        x = dev(a)
        got, idx, val, nexc = hip.bitpfr_encode(x, origin, nbits)
        eo, ei, ev = oracle.bitpfr_encode(a, origin, nbits)
        assert nexc == ei.size, nbits
        assert np.array_equal(host(got), eo), nbits
        assert np.array_equal(host(idx).astype(np.uint64), ei), nbits
        assert np.array_equal(host(val).astype(np.uint64), ev), nbits
        back = host(hip.bitpfr_decode(got, n, origin, nbits, idx, val, dt))
        assert np.array_equal(back, a), nbits


def test_sparse_exceptions_far_apart(hip, oracle):
    # exceptions thousands of tiles apart: the look-back for the previous one
    n = 3_000_000
    a = np.zeros(n, np.uint32)
    for p in (5, 1_234_567, 1_234_568, 2_999_999):
        a[p] = 1 << 20
    got, idx, val, nexc = hip.bitpfr_encode(dev(a), 0, 3)
    eo, ei, ev = oracle.bitpfr_encode(a, 0, 3)
    assert nexc == 4 and np.array_equal(host(idx).astype(np.uint64), ei)
    assert np.array_equal(host(val).astype(np.uint64), ev)
    assert np.array_equal(host(hip.bitpfr_decode(got, n, 0, 3, idx, val, np.uint32)), a)


@pytest.mark.parametrize("dt", [np.uint8, np.int16, np.uint32, np.int64])
def test_bitpfr_lists_mixed_density(hip, oracle, dt):
    # stretches with few exceptions, with exactly as many as a tile of the packer keeps
    # notes for, with more than that and with none at all, in one sequence; the source
    # also at an address that is no multiple of 16; a list capacity below the count
    dt = np.dtype(dt)
    W = dt.itemsize * 8
    rng = np.random.default_rng(21)
    n = 70_001
    nbits = 3
    a = rng.integers(0, 8, n).astype(dt)
    out = np.zeros(n, bool)
    out[:8192] = rng.random(8192) < 0.01          # sparse
    out[8192:16384:32] = True                      # one in 32: exactly the room of a tile
    out[16384:24576:3] = True                      # dense
    out[40_000:] = rng.random(n - 40_000) < 0.04   # around the limit, tile by tile
    a[out] |= dt.type(1 << (W - 2))
    eo, ei, ev = oracle.bitpfr_encode(a, 0, nbits)
    for shift in (0, 1):
        buf = dev(np.concatenate([np.zeros(shift, dt), a]))
        x = buf[shift:]
        got, idx, val, nexc = hip.bitpfr_encode(x, 0, nbits)
        assert nexc == ei.size
        assert np.array_equal(host(got), eo)
        assert np.array_equal(host(idx).astype(np.uint64), ei)
        assert np.array_equal(host(val).astype(np.uint64), ev)
        assert np.array_equal(host(hip.bitpfr_decode(got, n, 0, nbits, idx, val, dt)), a)
    got, idx, val, nexc = hip.bitpfr_encode(dev(a), 0, nbits, capacity=1000)
    assert nexc == ei.size
    assert np.array_equal(host(idx).astype(np.uint64)[:1000], ei[:1000])
    assert np.array_equal(host(val).astype(np.uint64)[:1000], ev[:1000])


@pytest.mark.parametrize("nblocks", [1, 7, 2047, 2048, 2049, 5000])
def test_offsets_table_at_any_8_byte_address(hip, oracle, nblocks):
    # the offsets scan writes 16 bytes at a time where the table's address allows and
    # 8 bytes at a time elsewhere; tables on either side of a tile of the scan (2048 blocks),
    # a short last block, and the table handed back to the decoder from the same address
    import torch

    n = nblocks * 256 - 56
    a = dg.outlier_mix(n, np.int32, seed=23, shift=-200)
    x = dev(a)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    enc = hip.GroupEncoder(x.dtype, n, x.device)
    for shift in (0, 1):
        table = torch.zeros(nblocks + 2, dtype=torch.int64, device=x.device)
        enc.offsets = table[shift:shift + nblocks + 1]
        assert enc.offsets.data_ptr() % 16 == 8 * shift
        total = enc.encode(x, want_total=True)
        assert total == exp.size
        assert np.array_equal(host(enc.offsets).astype(np.uint64), exp_offs)
        assert np.array_equal(host(enc.out[:total]), exp)
        assert np.array_equal(host(hip.group_decode(enc.out[:total], enc.offsets, n, np.int32)), a)


def test_headline_distribution_round_trip(hip, oracle):
    # the bench workload at a size the oracle still checks byte for byte
    n = 256 * 4096
    a = dg.outlier_mix(n, np.int32, seed=3, shift=-200)
    x = dev(a)
    stream, offs = hip.group_encode(x)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    assert np.array_equal(host(offs).astype(np.uint64), exp_offs)
    assert np.array_equal(host(stream), exp)
    assert np.array_equal(host(hip.group_decode(stream, offs, n, np.int32)), a)
    d = hip.group_select(x)
    assert (d["encoding"] == 6).mean() > 0.8


def test_large_round_trip_properties(hip):
    # 2^27 values: round trip, offsets are the running sum of block sizes,
    # encoding is deterministic
    import torch

    n = 1 << 27
    a = dg.outlier_mix(n, np.int32, seed=3, shift=-200)
    x = dev(a)
    s1, o1 = hip.group_encode(x)
    s2, o2 = hip.group_encode(x)
    assert torch.equal(o1, o2) and torch.equal(s1, s2)
    d = hip.group_select(x)
    sizes = d["size"].astype(np.uint64)
    assert np.array_equal(host(o1).astype(np.uint64)[1:], np.cumsum(sizes))
    back = hip.group_decode(s1, o1, n, np.int32)
    assert torch.equal(back, x)


def test_host_entry_points(hip, oracle):
    a = dg.outlier_mix(70001, np.uint64, seed=6, body_bits=20, out_bits=50, shift=1 << 40)
    stream, offs = hip.group_encode_host(a)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    assert np.array_equal(offs, exp_offs) and np.array_equal(stream, exp)
    assert np.array_equal(hip.group_decode_host(stream, offs, a.size, np.uint64), a)


def test_argument_errors(hip):
    import torch

    x = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(hip.OrochError):
        hip.bitpck_encode(x, 0)
    with pytest.raises(hip.OrochError):
        hip.bitpck_encode(x, 33)
    with pytest.raises(hip.OrochError):
        hip.bitpfr_encode(x, 0, 32)


@pytest.mark.parametrize("dt", dg.DTYPES)
def test_block_statistics_against_oracle(hip, oracle, dt):
    rng = np.random.default_rng(77 + dg.DTYPES.index(dt))
    W = np.dtype(dt).itemsize * 8
    blocks = [dg.fuzz_block(rng, dt, 256) for _ in range(200)]
    blocks.append(dg.fuzz_block(rng, dt, int(rng.integers(1, 256))))
    a = np.concatenate(blocks)
    st = hip.group_stats(dev(a))
    assert len(st) == len(blocks)
    for b, blk in enumerate(blocks):
        mn, mx, hist = oracle.stats(blk)
        assert int(st[b]["nvalues"]) == blk.size
        assert int(st[b]["min"]) == mn and int(st[b]["max"]) == mx, (b, st[b], mn, mx)
        assert np.array_equal(st[b]["histogram"][:W + 1].astype(np.uint64), hist[:W + 1]), b
        assert not st[b]["histogram"][W + 1:].any()


@pytest.mark.parametrize("dt", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32])
def test_four_block_decoder_paths(hip, oracle, dt):
    """Streams whose blocks share a width, so that the four-blocks-per-wavefront
    decoder takes its fast path: every width, frame of reference or zigzag,
    no exceptions / a few (varint lists) / many (bit-packed lists)."""
    import torch

    dt = np.dtype(dt)
    W = dt.itemsize * 8
    rng = np.random.default_rng(4242 + W + (dt.kind == "u"))
    u = np.dtype("u%d" % dt.itemsize)
    parts = []
    for nb in range(1, W + 1):
        for shape in range(6):
            nblocks = 8
            body = rng.integers(0, 1 << nb, size=nblocks * 256, dtype=np.uint64)
            body[::256] = (1 << nb) - 1          # the width is reached in every block
            body[1::256] = 0
            if shape in (2, 3, 4, 5) and nb < W:
                # exceptions: a few, a dozen, dozens; the last shape clusters them
                count = {2: 2, 3: 12, 4: 40, 5: 70}[shape]
                for b in range(nblocks):
                    pos = rng.choice(np.arange(2, 256), size=min(count, 200), replace=False)
                    if shape == 5:
                        pos = np.arange(2, 2 + count)
                    hi = rng.integers(1, 1 << (W - nb), size=pos.size, dtype=np.uint64)
                    body[b * 256 + pos] |= hi << np.uint64(nb)
            origin = 0 if shape == 0 else int(rng.integers(0, 1 << W))
            v = (body + np.uint64(origin)) & np.uint64((1 << W) - 1)
            parts.append(v.astype(u).view(dt))
    a = np.concatenate(parts)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    x = dev(a)
    stream, offs = hip.group_encode(x)
    assert np.array_equal(host(offs).astype(np.uint64), exp_offs)
    assert np.array_equal(host(stream), exp)
    encs = set(int(e) for e in hip.group_select(x)["encoding"])
    assert {4, 5, 6} & encs == {4, 5, 6} or W == 8, encs
    back = hip.group_decode(stream, offs, a.size, dt)
    assert np.array_equal(host(back).view(dt), a)
    # the one-block-per-wavefront decoder agrees
    with hip.option("decode_one_block", 1):
        back1 = hip.group_decode(stream, offs, a.size, dt)
    assert torch.equal(back1, back)
    # a stream that does not end on a four-block boundary, with a short last block
    for cut in (256 * 5 + 17, 256 * 2, 255, 1):
        s2, o2 = hip.group_encode(x[:cut])
        assert np.array_equal(host(hip.group_decode(s2, o2, cut, dt)).view(dt), a[:cut])


@pytest.mark.parametrize("dt", [np.uint8, np.int16, np.uint32, np.int32])
def test_streams_of_one_class_of_block(hip, oracle, dt):
    """Streams made of one class of block each, of those the four-block kernels
    treat apart: constant blocks, raw blocks, widths of 1 to 3 bits, more than 16
    exceptions with bit-packed lists, varint and varfor blocks.  Bytes against the
    oracle, and both decoders."""
    import torch

    dt = np.dtype(dt)
    W = dt.itemsize * 8
    u = np.dtype("u%d" % dt.itemsize)
    rng = np.random.default_rng(31 + W + (dt.kind == "u"))
    lo = -(1 << (W - 1)) if dt.kind == "i" else 0
    hi = (1 << (W - 1)) - 1 if dt.kind == "i" else (1 << W) - 1
    nblocks = 40

    def stream_of(kind):
        out = []
        for _ in range(nblocks):
            if kind == "constant":
                v = np.full(256, int(rng.integers(lo, hi + 1)), np.int64)
            elif kind == "raw":
                v = rng.integers(lo, hi + 1, 256).astype(np.int64)
            elif kind == "narrow":
                v = rng.integers(0, 1 << int(rng.integers(1, 4)), 256).astype(np.int64) + int(rng.integers(lo, hi - 8))
            elif kind == "dense":
                b = max(4, W // 2 - 2)
                v = rng.integers(0, 1 << b, 256).astype(np.int64)
                pos = rng.choice(256, int(rng.integers(17, 120)), replace=False)
                v[pos] = rng.integers(1 << b, 1 << (b + 3), pos.size)
            else:  # varint / varfor: seven-bit values and a few wide ones
                v = rng.integers(0, 1 << 7, 256).astype(np.int64)
                pos = rng.choice(256, int(rng.integers(1, 4)), replace=False)
                v[pos] = rng.integers(1 << (W - 2), 1 << (W - 1), pos.size)
                if kind == "varint" and dt.kind == "i":
                    v -= 1 << 6
                if kind == "varfor":
                    v = np.minimum(v + (1 << (W - 2)) + lo, hi)
            out.append(v)
        return np.concatenate(out).astype(u).view(dt) if dt.kind == "i" else np.concatenate(out).astype(u)

    # (for the narrower types a plain packing sometimes beats the patched one)
    want_encoding = {"constant": {0}, "raw": {1}, "narrow": {4, 5}, "dense": {6} if W == 32 else {4, 5, 6},
                     "varint": {2}, "varfor": {3}}
    for kind in ("constant", "raw", "narrow", "dense", "varint", "varfor"):
        if W == 8 and kind in ("varint", "varfor"):
            continue  # (no block of an 8-bit type is coded as varints)
        a = stream_of(kind)
        exp, exp_offs = oracle.blocks_encode(a, 256, 16)
        x = dev(a)
        desc = hip.group_select(x)
        encs = set(int(e) for e in desc["encoding"])
        assert encs <= want_encoding[kind], (kind, encs)
        if kind == "narrow":
            assert int(desc["nbits"].max()) <= 3
        if kind == "dense":
            patched = desc[desc["encoding"] == 6]
            assert W < 32 or len(patched) == nblocks
            assert len(patched) == 0 or int(patched["noutliers"].min()) > 16
        stream, offs = hip.group_encode(x)
        assert np.array_equal(host(offs).astype(np.uint64), exp_offs), kind
        assert np.array_equal(host(stream), exp), kind
        back = hip.group_decode(stream, offs, a.size, dt)
        assert np.array_equal(host(back).view(dt), a), kind
        with hip.option("decode_one_block", 1):
            assert torch.equal(hip.group_decode(stream, offs, a.size, dt), back), kind


@pytest.mark.parametrize("dt", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32])
def test_varint_blocks_within_quads(hip, oracle, dt):
    """Blocks coded as varints (zigzag images, or deltas from an origin) in every
    position of a four-block group, next to bit-packed and constant blocks: the
    four-block kernels take the varint blocks of a group in a pass of their own."""
    import torch

    dt = np.dtype(dt)
    W = dt.itemsize * 8
    u = np.dtype("u%d" % dt.itemsize)
    rng = np.random.default_rng(77 + W + (dt.kind == "u"))

    def block(kind):
        if kind in (0, 1):
            # seven-bit values and a few wide ones: a byte per value beats every packing
            # (integer_codec.h:234-384); moved away from zero, the deltas from the
            # smallest value are what is coded
            v = rng.integers(0, 1 << 7, 256).astype(np.int64)
            pos = rng.choice(256, int(rng.integers(1, 4)), replace=False)
            v[pos] = rng.integers(1 << (W - 2), 1 << (W - 1), pos.size)
            if kind == 0 and dt.kind == "i":
                v -= 1 << 6
            if kind == 1:
                v += (1 << (W - 2)) - ((1 << (W - 1)) if dt.kind == "i" else 0)
                v = np.minimum(v, (1 << (W - 1)) - 1 if dt.kind == "i" else (1 << W) - 1)
            return v
        if kind == 2:
            return rng.integers(0, 1 << min(W - 1, 9), 256).astype(np.int64)
        return np.full(256, int(rng.integers(0, 100)), np.int64)

    blocks = []
    for mask in range(16):
        for vk in (0, 1):
            blocks += [block(vk if (mask >> g) & 1 else 2 + (mask + g) % 2) for g in range(4)]
    blocks += [block(int(rng.integers(0, 4))) for _ in range(4 * 64)]
    a = np.concatenate(blocks).astype(np.int64).astype(u).view(dt) if dt.kind == "i" else np.concatenate(blocks).astype(u)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    x = dev(a)
    stream, offs = hip.group_encode(x)
    assert np.array_equal(host(offs).astype(np.uint64), exp_offs)
    assert np.array_equal(host(stream), exp)
    encs = [int(e) for e in hip.group_select(x)["encoding"]]
    # (no block of an 8-bit type is ever coded as varints: the rule prefers a packing)
    assert W == 8 or (encs.count(2) > 40 and encs.count(3) > 40), (encs.count(2), encs.count(3))
    back = hip.group_decode(stream, offs, a.size, dt)
    assert np.array_equal(host(back).view(dt), a)
    with hip.option("decode_one_block", 1):
        assert torch.equal(hip.group_decode(stream, offs, a.size, dt), back)
    # a varint block cut short is the last block of its stream
    for cut in (256 * 4 + 100, 256 * 7 + 1):
        s2, o2 = hip.group_encode(x[:cut])
        e2, eo2 = oracle.blocks_encode(a[:cut], 256, 16)
        assert np.array_equal(host(s2), e2)
        assert np.array_equal(host(hip.group_decode(s2, o2, cut, dt)).view(dt), a[:cut])


@pytest.mark.parametrize("dt", [np.int16, np.uint32, np.int64])
def test_serialised_container(hip, oracle, dt):
    import torch

    rng = np.random.default_rng(99)
    a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(300)] + [dg.fuzz_block(rng, dt, 77)])
    c = hip.container_encode(dev(a))
    h = hip.container_header(c)
    nb = 301
    assert h["magic"] == b"OROCHBK1" and int(h["nvalues"]) == a.size and int(h["nblocks"]) == nb
    assert int(h["table_offset"]) == 64 and int(h["stream_offset"]) == (64 + 8 * (nb + 1) + 15) // 16 * 16
    assert c.numel() == int(h["stream_offset"]) + int(h["stream_bytes"])
    raw = host(c)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    table = raw[64:64 + 8 * (nb + 1)].view(np.uint64)
    assert np.array_equal(table, exp_offs)
    assert not raw[64 + 8 * (nb + 1):int(h["stream_offset"])].any()
    assert np.array_equal(raw[int(h["stream_offset"]):], exp)
    assert np.array_equal(host(hip.container_decode(c)).view(dt), a)
    # any range of blocks on its own
    for first, count in ((0, 1), (5, 4), (17, 130), (300, 1), (299, 2), (42, 0)):
        got = host(hip.container_decode(c, first, count)).view(dt)
        assert np.array_equal(got, a[first * 256:(first + count) * 256]), (first, count)
    # host buffers on both sides: the same bytes
    ch = hip.container_encode_host(a)
    assert np.array_equal(ch, raw)
    assert np.array_equal(hip.container_decode_host(ch), a)
    assert np.array_equal(hip.container_decode_host(ch, 100, 3), a[25600:26368])
    # damaged or truncated containers are refused
    bad = ch.copy()
    bad[3] ^= 1
    with pytest.raises(hip.OrochError):
        hip.container_decode_host(bad)
    with pytest.raises(hip.OrochError):
        hip.container_decode_host(ch[:-16])
    with pytest.raises(hip.OrochError):
        hip.container_decode_host(ch, 301, 1)
    # an empty array still makes a valid container
    e = hip.container_encode_host(np.zeros(0, dt))
    assert e.size == 80 and hip.container_decode_host(e).size == 0


@pytest.mark.parametrize("dt", [np.int32, np.int64])
def test_damaged_streams_stay_inside_the_stream(hip, dt):
    """Decoding is for files and network buffers: a damaged offsets table or
    damaged block bytes may give wrong values, never a read outside the stream
    (which would show up here as a GPU fault)."""
    import torch

    rng = np.random.default_rng(5)
    a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(64)])
    stream, offs = hip.group_encode(dev(a))
    n = a.size
    good = host(hip.group_decode(stream, offs, n, dt))
    assert np.array_equal(good.view(dt), a)
    total = int(offs[-1])
    # offsets: out of order, beyond the end, misaligned, enormous (the last entry,
    # the stream length the caller vouches for, stays)
    o = host(offs).astype(np.uint64)
    bad = o.copy()
    bad[3], bad[4] = bad[4], bad[3]
    bad[10] = np.uint64(total + 4096)
    bad[17] += np.uint64(5)
    bad[30] = np.uint64(2 ** 63 + 12345)
    bad[40] = np.uint64(0)
    got = host(hip.group_decode(stream, torch.from_numpy(bad.astype(np.int64)).cuda(), n, dt)).view(dt)
    ok_blocks = [b for b in range(64) if all(bad[b + d] == o[b + d] for d in (0, 1))]
    for b in ok_blocks[:8] + ok_blocks[-8:]:
        assert np.array_equal(got[b * 256:(b + 1) * 256], a[b * 256:(b + 1) * 256]), b
    # block bytes: random damage in a copy of the stream
    s = host(stream).copy()
    hits = rng.integers(0, s.size, size=400)
    s[hits] ^= rng.integers(1, 256, size=400).astype(np.uint8)
    hip.group_decode(torch.from_numpy(s).cuda(), offs, n, dt)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dt", [np.int16, np.int32, np.int64])
def test_damaged_container_tables(hip, dt):
    """The offsets table of a container comes from a file or the network.  With
    the table in host memory a damaged entry is an error; with the container in
    device memory the decoder keeps every block inside the stream, so entries
    far beyond it (which would otherwise be a GPU fault) give wrong values at
    most; an array does not load a container with a damaged table."""
    import torch

    rng = np.random.default_rng(17)
    a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(40)] + [dg.fuzz_block(rng, dt, 100)])
    good = hip.container_encode_host(a)
    h = good[:64].view(hip.capi.HEADER_DTYPE)[0]
    nb, stream_bytes = int(h["nblocks"]), int(h["stream_bytes"])

    def damaged(entry, value):
        c = good.copy()
        c[64:64 + 8 * (nb + 1)].view(np.uint64)[entry] = np.uint64(value)
        return c

    cases = [(nb, 1 << 40), (nb, stream_bytes + 16), (7, 1 << 40), (7, 8), (12, 0), (nb - 1, 1 << 62), (0, 16)]
    for entry, value in cases:
        c = damaged(entry, value)
        with pytest.raises(hip.OrochError):
            hip.container_decode_host(c)
        arr = hip.IntegerArray(dt)
        arr.append(a[:300])
        with pytest.raises(hip.OrochError):
            arr.load(c)
        # the array is as it was
        assert len(arr) == 300 and np.array_equal(arr.read(0, 300), a[:300])
        # on the device nobody has looked at the table: no fault, and the blocks
        # whose entries (and those of the three they may share a wavefront with) are
        # intact come out right
        d = torch.from_numpy(c).cuda()
        got = host(hip.container_decode(d)).view(dt)
        torch.cuda.synchronize()
        assert got.size == a.size
        for b in range(nb):
            # the four-block decoder takes the extents of four blocks together
            if not 4 * (b // 4) <= entry <= 4 * (b // 4) + 4 and entry != nb:
                assert np.array_equal(got[b * 256:(b + 1) * 256], a[b * 256:(b + 1) * 256]), (entry, value, b)
        # a range that ends at the damaged entry: it is the limit the kernel reads
        if 0 < entry < nb:
            hip.container_decode(d, 0, entry)
            torch.cuda.synchronize()
    # blocks that do not touch the damaged entry decode through the host path too
    c = damaged(20, 1 << 40)
    assert np.array_equal(hip.container_decode_host(c, 0, 19), a[:19 * 256])
    assert np.array_equal(hip.container_decode_host(c, 21, nb - 21), a[21 * 256:])
    arr = hip.IntegerArray(dt)
    arr.load(good)
    assert len(arr) == a.size and np.array_equal(arr.read(0, a.size), a)


@pytest.mark.parametrize("dt", [np.uint8, np.int16, np.uint32, np.int64])
def test_batched_fetch(hip, oracle, dt):
    """Random access into packed sequences (bitpck.h:258-266, bitfor.h:71-74)."""
    import torch

    dt = np.dtype(dt)
    W = dt.itemsize * 8
    rng = np.random.default_rng(31)
    n = 50021
    for nbits in sorted({1, 3, 7, W // 2 + 1, W - 1, W}):
        a = dg.uniform_bits(n, dt, nbits, min(nbits, W - 1) if dt.kind == "i" else nbits)
        if dt.kind == "i":
            a = (a >> 1) * np.where(a & 1, -1, 1).astype(dt)   # zigzag image below 2^nbits
        idx = rng.integers(0, n, size=20000).astype(np.int64)
        idx[:3] = (0, n - 1, n // 2)
        b = dev(oracle.bitpck_encode(a, nbits))
        got = host(hip.bitpck_fetch(b, n, nbits, dev(idx), dt)).view(dt)
        assert np.array_equal(got, a[idx]), nbits
        if nbits < W:
            origin = int(rng.integers(-1000, 1000)) if dt.kind == "i" else int(rng.integers(0, 100))
            f = (dg.uniform_bits(n, np.dtype("u%d" % dt.itemsize), nbits + 1, nbits).astype(np.int64) + origin)
            f = f.astype(dt)
            b = dev(oracle.bitfor_encode(f, origin, nbits))
            got = host(hip.bitpck_fetch(b, n, nbits, dev(idx), dt, origin=origin)).view(dt)
            assert np.array_equal(got, f[idx]), nbits


@pytest.mark.parametrize("dt", dg.DTYPES)
def test_varint_sequences(hip, oracle, golden, dt):
    """varint_codec<T>::space / encode / decode over whole sequences (varint.h:124-147)."""
    import torch

    dt = np.dtype(dt)
    W = dt.itemsize * 8
    # the golden cases of this element type
    z, cases, fields = golden
    for i, c in enumerate(cases):
        if c["kind"] != "varint" or dg.DTYPES[c["dtype"]] != dt.type:
            continue
        a, exp = z["c%d_in" % i].view(dt), z["c%d_out" % i]
        if a.size == 0:
            continue
        got = host(hip.varint_encode(dev(a)))
        assert np.array_equal(got, exp), c
        assert np.array_equal(host(hip.varint_decode(dev(exp), a.size, dt)).view(dt), a), c
    # every length of value, across tile boundaries, ragged sizes
    rng = np.random.default_rng(17 + W)
    for n in (1, 15, 4095, 4096, 4097, 100003):
        bits = rng.integers(0, W + 1, size=n)
        raw = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
        mask = np.where(bits >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << bits.astype(np.uint64)) - np.uint64(1))
        a = (raw & mask).astype(np.dtype("u%d" % dt.itemsize)).view(dt)
        exp = oracle.varint_encode(a)
        x = dev(a)
        assert hip.varint_space(x) == exp.size
        got = hip.varint_encode(x)
        assert np.array_equal(host(got), exp), n
        assert np.array_equal(host(hip.varint_decode(got.clone(), n, dt)).view(dt), a), n
    # tests/unit/varint.cc:20-44: known bytes
    known = {0: [0], 1: [1], 127: [0x7f], 128: [0x80, 1], 300: [0xac, 2]}
    if dt.kind == "u" and W >= 16:
        for v, b in known.items():
            assert host(hip.varint_encode(dev(np.array([v], dt)))).tolist() == b


@pytest.mark.parametrize("dt", [np.int8, np.int16, np.int32, np.int64])
def test_value_arrays_at_odd_offsets(hip, oracle, dt):
    """Values that start 4 bytes (8 for 64-bit elements) into an allocation: the
    least alignment the entry points take; less than that is refused."""
    import torch

    dt = np.dtype(dt)
    step = max(4, dt.itemsize) // dt.itemsize
    rng = np.random.default_rng(808 + dt.itemsize)
    a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(21)] + [dg.fuzz_block(rng, dt, 99)])
    whole = dev(np.concatenate([np.zeros(step, dt), a]))
    x = whole[step:]
    assert x.data_ptr() % 16 != 0
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    stream, offs = hip.group_encode(x)
    assert np.array_equal(host(offs).astype(np.uint64), exp_offs)
    assert np.array_equal(host(stream), exp)
    out = torch.zeros(a.size + step, dtype=x.dtype, device="cuda")
    code = hip.dtype_code(dt)
    rc = hip.lib.oroch_group_decode(code, stream.data_ptr(), offs.data_ptr(), a.size,
                                    out[step:].data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(host(out[step:]).view(dt), a)
    assert not host(out[:step]).any()
    if dt.itemsize < 4:
        y = whole[1:]
        desc = torch.empty(32 * 32, dtype=torch.uint8, device="cuda")
        assert hip.lib.oroch_group_select(code, y.data_ptr(), 256, desc.data_ptr(), None) == -5  # OROCH_E_ARGUMENT

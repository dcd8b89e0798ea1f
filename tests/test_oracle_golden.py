"""The CPU oracle against the reference's golden vectors and known answers.

Expected outputs in tests/golden/golden_v1.npz were produced by the reference's
own headers (tests/golden/make_golden.py).  The known-answer cases restate the
assertions of the reference's unit tests (cited per test).
"""
import numpy as np
import pytest

import datagen as dg
from oracle import pyoracle as po


def _input(z, i, c):
    return z["c%d_in" % i].view(dg.DTYPES[c["dtype"]])


def test_golden_groups(golden, oracle):
    z, cases, fields = golden
    n = 0
    for i, c in enumerate(cases):
        if c["kind"] != "group":
            continue
        a = _input(z, i, c)
        m = oracle.select(a)
        assert list(m.astuple()) == c["meta"], (i, c)
        exp = z["c%d_out" % i]
        got = oracle.group_encode(a)
        assert np.array_equal(got, exp), (i, c)
        assert np.array_equal(oracle.group_decode(exp, a.size, a.dtype), a), (i, c)
        n += 1
    assert n > 900


def test_golden_blocks(golden, oracle):
    z, cases, _ = golden
    for i, c in enumerate(cases):
        if c["kind"] != "blocks":
            continue
        a = _input(z, i, c)
        out, offs = oracle.blocks_encode(a, c["block"], c["align"])
        assert np.array_equal(offs, z["c%d_offs" % i])
        assert np.array_equal(out, z["c%d_out" % i])
        assert np.array_equal(oracle.blocks_decode(out, offs, a.size, a.dtype, c["block"]), a)


def test_golden_whole_sequence(golden, oracle):
    z, cases, _ = golden
    seen = set()
    for i, c in enumerate(cases):
        k = c["kind"]
        if k not in ("bitpck", "bitfor", "bitpfr", "varint"):
            continue
        seen.add(k)
        a = _input(z, i, c)
        exp = z["c%d_out" % i]
        dt = a.dtype
        if k == "bitpck":
            assert np.array_equal(oracle.bitpck_encode(a, c["nbits"]), exp), c
            back = oracle.bitpck_decode(exp, a.size, c["nbits"], dt)
            W = dt.itemsize * 8
            if c["nbits"] >= W:  # narrower widths truncate (bitblk.cc:92-94)
                assert np.array_equal(back, a), c
        elif k == "bitfor":
            origin = np.array([c["origin"]], np.uint64).astype("u%d" % dt.itemsize).view(dt)[0]
            assert np.array_equal(oracle.bitfor_encode(a, origin, c["nbits"]), exp), c
        elif k == "bitpfr":
            origin = np.array([c["origin"]], np.uint64).astype("u%d" % dt.itemsize).view(dt)[0]
            out, idx, val = oracle.bitpfr_encode(a, origin, c["nbits"])
            assert np.array_equal(out, exp), c
            assert np.array_equal(idx, z["c%d_idx" % i]), c
            assert np.array_equal(val, z["c%d_val" % i]), c
            back = oracle.bitpfr_decode(out, a.size, origin, c["nbits"], idx, val, dt)
            assert np.array_equal(back, a), c
        else:
            assert np.array_equal(oracle.varint_encode(a), exp), c
            assert np.array_equal(oracle.varint_decode(exp, a.size, dt), a), c
    assert seen == {"bitpck", "bitfor", "bitpfr", "varint"}


# ---- known answers held by the reference's own unit tests -------------------

def test_varint_known_answers(oracle):
    # tests/unit/varint.cc:9-18 (sizes) and :20-44 (bytes)
    sizes = {0: 1, 1: 1, 127: 1, 128: 2, 16383: 2, 16384: 3, 0xFFFFFFFFFFFFFFFF: 10}
    for v, s in sizes.items():
        assert oracle.varint_encode(np.array([v], np.uint64)).size == s
    enc = lambda v: list(oracle.varint_encode(np.array([v], np.uint32)))
    assert enc(1) == [1] and enc(2) == [2] and enc(127) == [127]
    assert enc(128) == [0x80, 0x01]


def test_zigzag_known_answers(oracle):
    # tests/unit/zigzag.cc:11-15, 38-42: 0,-1,1,-2,2 -> 0,1,2,3,4 (seen through
    # single-byte varints of signed values)
    for dt in (np.int8, np.int16, np.int32, np.int64):
        got = oracle.varint_encode(np.array([0, -1, 1, -2, 2], dt))
        assert list(got) == [0, 1, 2, 3, 4]
        info = np.iinfo(dt)
        ext = np.array([info.min, info.max], dt)
        assert np.array_equal(oracle.varint_decode(oracle.varint_encode(ext), 2, dt), ext)


def test_bitpfr_exception_bookkeeping(oracle):
    # tests/unit/bitpfr.cc:13-30: FREF then FREF + (1 << 7) -> one exception,
    # stored index 1, stored value 1
    a = np.array([1000, 1000 + (1 << 7)], np.uint32)
    _, idx, val = oracle.bitpfr_encode(a, 1000, 7)
    assert list(idx) == [1] and list(val) == [1]
    # tests/unit/offset.cc:10-32: positions 1,2,3,5 -> gaps 1,0,0,1
    b = np.zeros(8, np.uint32)
    b[[1, 2, 3, 5]] = 1 << 9
    _, idx, _ = oracle.bitpfr_encode(b, 0, 4)
    assert list(idx) == [1, 0, 0, 1]
    # tests/unit/bitpfr.cc:32-61: 128 small + 24 outliers
    c = np.concatenate([np.arange(128) + 1000, 1000 + (1 << (7 + np.arange(24)))]).astype(np.uint32)
    out, idx, val = oracle.bitpfr_encode(c, 1000, 7)
    assert idx.size == 24 and val.size == 24
    assert np.array_equal(oracle.bitpfr_decode(out, c.size, 1000, 7, idx, val, np.uint32), c)


def test_worked_example_known_answers(oracle):
    # int32 ramp 3i-100 with one 2^30 spike: metadata bytes and data size
    a = (3 * np.arange(256) - 100).astype(np.int32)
    a[17] = 1 << 30
    m = oracle.select(a)
    assert m.encoding == po.BITPFR
    assert list(oracle.meta_encode(m, np.int32)) == [0x06, 0xC7, 0x01, 0x0A, 0x01, 0x00, 0x00]
    data = oracle.codec_encode(a, m)
    assert data.size == 356 and list(data[-4:]) == [0x11, 0x80, 0x80, 0x40]
    # stand-alone bitpfr<int32>(origin -50, nbits 6)
    b = np.array([-50, -49, 14, 1000, -50, 5000, 5001, 0], np.int32)
    out, idx, val = oracle.bitpfr_encode(b, -50, 6)
    assert list(idx) == [2, 0, 1, 0] and list(val) == [1, 16, 78, 78]
    low = oracle.bitfor_decode(out, 8, 0, 6, np.uint32)
    assert list(low) == [0, 1, 0, 26, 0, 58, 59, 50]


def test_bitpck_masks_high_bits(oracle):
    # tests/unit/bitblk.cc:25-27, 92-94: values are truncated to nbits
    a = np.array([0xFFFFFFFF, 5, 0x12345678], np.uint32)
    back = oracle.bitpck_decode(oracle.bitpck_encode(a, 4), 3, 4, np.uint32)
    assert list(back) == [0xF, 5, 8]


def test_integer_codec_every_length(oracle):
    # tests/unit/integer_codec.cc:8-43: ramp 0..n-1 for every n in 0..128
    for n in range(129):
        a = np.arange(n, dtype=np.uint32)
        g = oracle.group_encode(a)
        assert np.array_equal(oracle.group_decode(g, n, np.uint32), a)


@pytest.mark.skipif(po.ref is None, reason="reference build (oracle/_ref) not present")
def test_oracle_matches_reference_fuzz(oracle):
    rng = np.random.default_rng(7)
    encs = set()
    for it in range(4000):
        dt = dg.DTYPES[it % 8]
        n = int(rng.integers(0, 300)) if it % 2 else 256
        a = dg.fuzz_block(rng, dt, n)
        mo, mr = oracle.select(a), po.ref.select(a)
        assert mo.astuple() == mr.astuple(), (dt, n)
        encs.add(mo.encoding)
        assert np.array_equal(oracle.group_encode(a), po.ref.group_encode(a))
    assert encs == set(range(7))

"""One group of any length through the device path: integer_codec<T>::select /
encode / decode and integer_group<T> on sequences longer than a block
(integer_codec.h:234-402, integer_group.h:44-76), against the oracle, which
takes any n."""
import numpy as np
import pytest

import datagen as dg

pytestmark = pytest.mark.gpu

NAUGHT, NORMAL, VARINT, VARFOR, BITPCK, BITFOR, BITPFR = range(7)


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


def shapes(dt, n, seed):
    """Sequences of n values that between them reach every encoding."""
    dt = np.dtype(dt)
    W = dt.itemsize * 8
    rng = np.random.default_rng(seed)
    signed = dt.kind == "i"
    out = {
        "outliers": dg.outlier_mix(n, dt, seed=seed, body_bits=min(10, W - 3), out_bits=min(28, W - 1),
                                   shift=-100 if signed else 0),
        "narrow": dg.uniform_bits(n, dt, seed, max(1, W // 3), shift=5),
        "constant": np.full(n, 77, dt),
        "full": dg.uniform_bits(n, dt, seed + 1, W),
        "ramp": (np.arange(n, dtype=np.uint64) * np.uint64(3)).astype(np.dtype("u%d" % dt.itemsize)).view(dt),
    }
    # mostly tiny with some medium values: the varint family
    tiny = rng.integers(0, 8, size=n).astype(np.uint64)
    some = rng.random(n) < 0.2
    tiny[some] = rng.integers(0, 1 << min(W - 1, 20), size=int(some.sum())).astype(np.uint64)
    out["tiny"] = tiny.astype(np.dtype("u%d" % dt.itemsize)).view(dt)
    if signed:
        out["around_zero"] = (rng.integers(-40, 40, size=n)).astype(dt)
    # many outliers of assorted widths, dense enough for packed lists
    many = rng.integers(0, 1 << max(1, W // 4), size=n).astype(np.uint64)
    hit = rng.random(n) < 0.3
    many[hit] = rng.integers(0, 1 << (W - 1), size=int(hit.sum()), dtype=np.uint64)
    out["dense_outliers"] = (many + np.uint64(1000 if W > 16 else 3)).astype(np.dtype("u%d" % dt.itemsize)).view(dt)
    return out


def desc_fields(d):
    r = d[0]
    return (int(r["encoding"]), int(r["nbits"]), int(r["origin"]), int(r["dataspace"]), int(r["metaspace"]),
            int(r["noutliers"]), int(r["index_encoding"]), int(r["index_nbits"]), int(r["value_encoding"]),
            int(r["value_nbits"]))


@pytest.mark.parametrize("n", [257, 1000, 4096, 100003])
@pytest.mark.parametrize("dt", dg.DTYPES)
def test_long_group_matches_oracle(hip, oracle, dt, n):
    seen = set()
    for name, a in shapes(dt, n, seed=n % 97 + np.dtype(dt).itemsize).items():
        x = dev(a)
        m = oracle.select(a)
        d = hip.sequence_select(x)
        assert desc_fields(d) == m.astuple(), (name, m)
        seen.add(m.encoding)
        img = hip.sequence_group_image(x)
        exp = oracle.group_encode(a)
        assert np.array_equal(img, exp), name
        # and back, from the image alone
        d2, metabytes = hip.sequence_meta_decode(img, dt)
        at = (metabytes + 7) & ~7
        back, used = hip.sequence_decode(dev(img[at:]) if img.size > at else dev(np.zeros(16, np.uint8)), n, d2, dt)
        assert np.array_equal(back.cpu().numpy().view(dt), a), name
        assert used == img.size - at, name
    assert BITPFR in seen and NAUGHT in seen


def test_group_bench_scenario(hip, oracle):
    # tests/bench/group-bench.cc:8-30: one group of 1000 values 0..999 (bitfor or bitpck)
    a = np.arange(1000, dtype=np.int32)
    x = dev(a)
    d = hip.sequence_select(x)
    assert desc_fields(d) == oracle.select(a).astuple()
    img = hip.sequence_group_image(x)
    assert np.array_equal(img, oracle.group_encode(a))
    d2, mb = hip.sequence_meta_decode(img, np.int32)
    back, _ = hip.sequence_decode(dev(img[(mb + 7) & ~7:]), 1000, d2, np.int32)
    assert np.array_equal(back.cpu().numpy(), a)


@pytest.mark.parametrize("dt", [np.int8, np.uint16, np.int32, np.uint32, np.int64])
@pytest.mark.parametrize("n", [100, 256, 5000])
def test_encode_honours_the_callers_metadata(hip, oracle, dt, n):
    """integer_codec<T>::encode takes whatever metadata it is handed
    (integer_codec.h:386-393): every encoding forced on the same values gives
    the bytes the oracle gives for that metadata, and decodes back."""
    from oracle import pyoracle as po

    dt = np.dtype(dt)
    W = dt.itemsize * 8
    mask = (1 << W) - 1
    a = dg.outlier_mix(n, dt, seed=5, body_bits=min(9, W - 4), out_bits=W - 2, permille=50,
                       shift=-20 if dt.kind == "i" else 7)
    x = dev(a)
    raw = a.view(np.dtype("u%d" % dt.itemsize)).astype(np.uint64)
    mn = int(a.min()) & mask
    rng_bits = int(int(a.max()) - int(a.min())).bit_length()

    def both(**kw):
        m = po.Meta()
        d = np.zeros(1, hip.capi.SEQUENCE_DESC)
        for k, v in kw.items():
            setattr(m, k, v)
            d[k] = v
        d["nvalues"] = n
        return m, d

    cases = [both(encoding=NORMAL), both(encoding=VARINT), both(encoding=VARFOR, origin=mn),
             both(encoding=BITPCK, nbits=W), both(encoding=BITFOR, origin=mn, nbits=rng_bits),
             both(encoding=BITFOR, origin=mn, nbits=min(W, rng_bits + 3))]
    for nbits in (3, min(9, W - 4), W - 3):
        u = (raw - np.uint64(mn)) & np.uint64(mask)
        out_pos = np.nonzero(u >> np.uint64(nbits))[0]
        k = out_pos.size
        if k == 0:
            continue
        gaps = np.diff(np.concatenate([[-1], out_pos])) - 1
        inb = max(1, int(gaps.max()).bit_length())
        vnb = int((u >> np.uint64(nbits)).max()).bit_length()
        for ie, ve in ((VARINT, VARINT), (BITPCK, BITPCK), (BITPCK, VARINT), (VARINT, BITPCK)):
            cases.append(both(encoding=BITPFR, origin=mn, nbits=nbits, noutliers=k, index_encoding=ie,
                              index_nbits=inb if ie == BITPCK else 0, value_encoding=ve,
                              value_nbits=vnb if ve == BITPCK else 0))
    for m, d in cases:
        exp = oracle.codec_encode(a, m)
        got = hip.sequence_encode(x, d, capacity=exp.size + 64).cpu().numpy()
        assert np.array_equal(got, exp), m
        src = dev(np.concatenate([got, np.zeros(16, np.uint8)]))[:got.size] if got.size else dev(np.zeros(16, np.uint8))[:0]
        back, used = hip.sequence_decode(src, n, d, dt)
        assert np.array_equal(back.cpu().numpy().view(dt), a), m
        assert used == exp.size, m
    # a bitpfr descriptor whose outlier count is not that of the values is refused
    m, d = both(encoding=BITPFR, origin=mn, nbits=3, noutliers=1, index_encoding=VARINT, value_encoding=VARINT)
    with pytest.raises(hip.OrochError):
        hip.sequence_encode(x, d, capacity=n * 16)
    # exception lists left at 'normal' (a descriptor nobody filled in) are refused: the
    # reference would write raw words there, a form no header announces
    u = (raw - np.uint64(mn)) & np.uint64(mask)
    k = int(np.count_nonzero(u >> np.uint64(3)))
    if k:
        for ie, ve in ((NORMAL, VARINT), (VARINT, NORMAL), (NORMAL, NORMAL)):
            m, d = both(encoding=BITPFR, origin=mn, nbits=3, noutliers=k, index_encoding=ie, value_encoding=ve)
            with pytest.raises(hip.OrochError):
                hip.sequence_encode(x, d, capacity=n * 16)
            with pytest.raises(hip.OrochError):
                hip.sequence_decode(dev(np.zeros(n * 16, np.uint8)), n, d, dt)
    # and too little room is an error, not an overrun
    m, d = both(encoding=NORMAL)
    with pytest.raises(hip.OrochError):
        hip.sequence_encode(x, d, capacity=n * dt.itemsize - 16)


def test_large_sequence_round_trip(hip, oracle):
    n = (1 << 22) + 12345
    a = dg.outlier_mix(n, np.int32, seed=9, shift=-200)
    x = dev(a)
    d = hip.sequence_select(x)
    assert desc_fields(d) == oracle.select(a).astuple()
    data = hip.sequence_encode(x, d)
    assert data.numel() == int(d["dataspace"][0])
    back, used = hip.sequence_decode(data, n, d, np.int32)
    assert used == data.numel() and np.array_equal(back.cpu().numpy(), a)

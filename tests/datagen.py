"""Deterministic synthetic inputs shared by the tests, the fixtures and bench.py.

`splitmix64(seed + i)` drives everything, so the same array can be produced on
any machine (and, index by index, on the device).
"""
import numpy as np

DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def h(seed, n, start=0):
    i = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return splitmix64(i + np.uint64(seed))


def _cast(u64, dt):
    """Wrap a uint64 array into dtype dt (two's complement truncation)."""
    dt = np.dtype(dt)
    u = np.dtype("u%d" % dt.itemsize)
    return u64.astype(u).view(dt)


def outlier_mix(n, dt, seed, body_bits=10, out_bits=28, permille=10, shift=0, start=0):
    """`body_bits`-bit noise with `permille`/1000 outliers of `out_bits` bits, plus `shift`.

    With the defaults and dt=uint32 this is the 1 % outlier distribution of the
    256M-value bitpfr configuration; dt=int32, shift=-200 is the headline one.
    """
    r = h(seed, n, start)
    v = r & np.uint64((1 << body_bits) - 1)
    sel = ((r >> np.uint64(32)) % np.uint64(1000)) < np.uint64(permille)
    big = ((r >> np.uint64(12)) | np.uint64(1 << (out_bits - 1))) & np.uint64((1 << out_bits) - 1)
    v = np.where(sel, v | big, v)
    with np.errstate(over="ignore"):
        v = v + np.uint64(shift & 0xFFFFFFFFFFFFFFFF)
    return _cast(v, dt)


def uniform_bits(n, dt, seed, bits, shift=0, start=0):
    r = h(seed, n, start)
    v = r & np.uint64((1 << bits) - 1) if bits < 64 else r
    with np.errstate(over="ignore"):
        v = v + np.uint64(shift & 0xFFFFFFFFFFFFFFFF)
    return _cast(v, dt)


def fuzz_block(rng, dt, n):
    """One random block drawn from a family of shapes that reaches every encoding."""
    dt = np.dtype(dt)
    W = dt.itemsize * 8
    u = np.dtype("u%d" % dt.itemsize)
    kind = rng.integers(0, 10)
    full = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(
        0, 2, size=n, dtype=np.uint64)
    if kind == 0:  # constant
        v = np.full(n, full[0] if n else 0, np.uint64)
    elif kind == 1:  # narrow noise around a random origin
        b = int(rng.integers(1, W + 1))
        v = (full >> np.uint64(64 - b)) + (full[0] if n else np.uint64(0))
    elif kind == 2:  # narrow noise plus a few wide outliers
        b = int(rng.integers(1, max(2, W - 2)))
        ob = int(rng.integers(b + 1, W + 1))
        v = full >> np.uint64(64 - b)
        k = int(rng.integers(0, max(1, n // 8) + 1))
        if n and k:
            pos = rng.integers(0, n, size=k)
            v[pos] = full[pos] >> np.uint64(64 - ob)
        v = v + (np.uint64(rng.integers(0, 1 << 20)) if rng.integers(0, 2) else np.uint64(0))
    elif kind == 3:  # small signed values around zero (varint friendly)
        b = int(rng.integers(1, 15))
        v = (full >> np.uint64(64 - b)) - np.uint64(1 << (b - 1))
    elif kind == 4:  # mostly tiny, some medium (varint / varfor)
        v = np.where(full % np.uint64(5) == 0, full >> np.uint64(64 - min(W, 20)),
                     full >> np.uint64(61))
        v = v + (np.uint64(rng.integers(0, 1 << 30)) if rng.integers(0, 2) else np.uint64(0))
    elif kind == 5:  # full range
        v = full
    elif kind == 6:  # ramp
        step = int(rng.integers(1, 1000))
        v = np.arange(n, dtype=np.uint64) * np.uint64(step) + np.uint64(rng.integers(0, 1 << 40))
    elif kind == 7:  # many outliers of assorted widths
        b = int(rng.integers(1, max(2, W // 2)))
        v = full >> np.uint64(64 - b)
        wide = full >> (np.uint64(64) - (full % np.uint64(W) + np.uint64(1)))
        v = np.where(full % np.uint64(3) == 0, wide, v)
    elif kind == 8:  # extremes of the type
        lo = np.uint64(1 << (W - 1)) if dt.kind == "i" else np.uint64(0)
        hi = np.uint64((1 << (W - 1)) - 1) if dt.kind == "i" else np.uint64((1 << W) - 1)
        v = np.where(full % np.uint64(7) == 0, lo, np.where(full % np.uint64(7) == 1, hi,
                                                           full >> np.uint64(60)))
    else:  # two clusters
        b = int(rng.integers(1, 12))
        v = (full >> np.uint64(64 - b)) + np.where(full % np.uint64(2) == 0, np.uint64(0),
                                                   np.uint64(1) << np.uint64(W - 2))
    with np.errstate(over="ignore"):
        return np.ascontiguousarray(v.astype(np.uint64).astype(u).view(dt))


def block_mix_int64(n, seed, block=256, start=0):
    """devgen.block_mix_int64 on the host: eight block shapes chosen by block id
    mod 8 (constant, 12-bit uniform, 12-bit with 1 % 40-bit outliers, signed with
    geometric magnitudes, large origin with small deltas, full 64-bit random,
    ramp, 3-bit)."""
    U = np.uint64
    r = h(seed, n, start)
    i = np.arange(start, start + n, dtype=np.uint64)
    blk = i // U(block)
    kind = blk % U(8)
    with np.errstate(over="ignore"):
        const = ((blk * U(0x9E3779B97F4A7C15)).view(np.int64) >> 7).view(np.uint64)
        u12 = r & U(0xfff)
        sel = ((r >> U(32)) % U(100)) == 0
        out40 = np.where(sel, u12 | (((r >> U(12)) | U(1 << 39)) & U((1 << 40) - 1)), u12)
        k = (r & U(63)) % U(36)
        mag = (r >> U(8)) & ((U(1) << k) - U(1))
        small = np.where((r & U(64)) != 0, U(0) - mag, mag)
        far = U(1 << 50) + (r & U(0xfff))
        ramp = (i % U(block)) * U(3) + blk
    v = const
    for kk, cand in ((1, u12), (2, out40), (3, small), (4, far), (5, r), (6, ramp), (7, r & U(7))):
        v = np.where(kind == U(kk), cand, v)
    return v.view(np.int64)

"""Device-side (torch) twins of datagen's splitmix64 generators, for inputs too
large to build on the host.  int64 arithmetic wraps like uint64; logical right
shifts are emulated."""
import torch

M64 = (1 << 64) - 1


def _s64(c):
    c &= M64
    return c - (1 << 64) if c >= (1 << 63) else c


def lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)


def h(seed, n, start, device):
    """splitmix64(seed + i) for i in [start, start + n) as int64 bit patterns."""
    i = torch.arange(start, start + n, dtype=torch.int64, device=device)
    z = i + _s64(seed + 0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def chunks(n, fn, dtype, device, chunk=1 << 26, start=0):
    """fn(first index, count) for pieces of [start, start + n), written into one tensor."""
    out = torch.empty(n, dtype=dtype, device=device)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        out[lo:lo + m] = fn(start + lo, m).to(dtype)
    return out


def uniform_bits(n, dtype, seed, bits, device, shift=0):
    return chunks(n, lambda lo, m: (h(seed, m, lo, device) & ((1 << bits) - 1)) + shift, dtype, device)


def outlier_mix(n, dtype, seed, device, body_bits=10, out_bits=28, permille=10, shift=0):
    def fn(lo, m):
        r = h(seed, m, lo, device)
        v = r & ((1 << body_bits) - 1)
        sel = (lsr(r, 32) % 1000) < permille
        big = (lsr(r, 12) | (1 << (out_bits - 1))) & ((1 << out_bits) - 1)
        return torch.where(sel, v | big, v) + shift
    return chunks(n, fn, dtype, device)


def block_mix_int64(n, seed, device, block=256, start=0):
    """Eight block shapes chosen by block id mod 8: constant, 12-bit uniform,
    12-bit with 1 % 40-bit outliers, signed with geometric magnitudes, large origin with
    small deltas, full 64-bit random, ramp, 3-bit.  Values [start, start + n) of the
    endless sequence (start a multiple of block)."""
    def fn(lo, m):
        r = h(seed, m, lo, device)
        i = torch.arange(lo, lo + m, dtype=torch.int64, device=device)
        blk = i // block
        kind = blk % 8
        const = (blk * _s64(0x9E3779B97F4A7C15)) >> 7
        u12 = r & 0xfff
        sel = (lsr(r, 32) % 100) == 0
        out40 = torch.where(sel, u12 | ((lsr(r, 12) | (1 << 39)) & ((1 << 40) - 1)), u12)
        k = (r & 63) % 36
        mag = lsr(r, 8) & (torch.bitwise_left_shift(torch.ones_like(r), k) - 1)
        small = torch.where((r & 64) != 0, -mag, mag)
        far = (1 << 50) + (r & 0xfff)
        ramp = (i % block) * 3 + blk
        v = const
        for k, cand in ((1, u12), (2, out40), (3, small), (4, far), (5, r), (6, ramp), (7, r & 7)):
            v = torch.where(kind == k, cand, v)
        return v
    return chunks(n, fn, torch.int64, device, chunk=1 << 25, start=start)

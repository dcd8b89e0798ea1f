"""CPU-side checks of the shipped library: it loads, exports every symbol the
header declares, and its host-only size arithmetic matches the oracle."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes

    import oroch_amd

    hdr = open(os.path.join(ROOT, "include", "oroch_hip.h")).read()
    declared = set(re.findall(r"\b(oroch_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = ctypes.CDLL(oroch_amd.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(oroch_amd.capi.EXPORTS)


def test_size_arithmetic_matches_oracle(oracle):
    import oroch_amd

    for nbits in range(1, 65):
        assert oroch_amd.lib.oroch_bitpck_capacity(nbits) == 128 // nbits
        for n in (0, 1, 2, 127, 128, 129, 1000, 1 << 20, (1 << 30) + 7):
            assert oroch_amd.bitpck_space(n, nbits) == oracle.pck_space(n, nbits)
    # BASELINE config 1: 2^20 uint32 at 13 bits
    assert oroch_amd.bitpck_space(1 << 20, 13) == 1864144
    for dt in oroch_amd.DTYPES:
        for n in (0, 1, 255, 256, 257, 100000):
            nb = (n + 255) // 256
            assert oroch_amd.group_bound(dt, n) >= nb * (8 + 256 * np.dtype(dt).itemsize)


def test_division_magic_is_exact():
    # floor(x / cap) == (x * ceil(2^32 / cap)) >> 32 for every x the kernels use
    for nbits in range(1, 65):
        cap = 128 // nbits
        magic = ((1 << 32) + cap - 1) // cap
        xs = np.arange(0, 1 << 16, dtype=np.uint64)
        assert np.array_equal((xs * np.uint64(magic)) >> np.uint64(32), xs // np.uint64(cap))
        for x in (1 << 20, (1 << 25) - 1):
            assert (x * magic) >> 32 == x // cap


def test_argument_errors_without_gpu():
    import oroch_amd

    L = oroch_amd.lib
    assert L.oroch_bitpck_encode(99, None, 1, 3, None, None) == -1      # dtype
    assert L.oroch_bitpck_encode(4, None, 1, 0, None, None) == -2       # nbits
    assert L.oroch_bitpck_encode(4, None, 1, 33, None, None) == -2
    assert L.oroch_bitpck_encode(4, None, 1, 3, None, None) == -5       # null pointers
    assert L.oroch_bitpck_encode(4, None, 0, 3, None, None) == 0        # empty is a no-op
    assert b"nbits" in L.oroch_hip_strerror(-2)


def test_container_check_refuses_impossible_headers():
    """A header is checked before anything is read through it: counts near 2^64
    must not wrap into a small number of blocks, and the table and the stream
    must fit the bytes that are there (host arithmetic only)."""
    import oroch_amd

    L = oroch_amd.lib

    def header(nvalues, nblocks=None, stream_bytes=0, stream_offset=None, dtype=4):
        if nblocks is None:
            nblocks = nvalues // 256 + (nvalues % 256 != 0)
        if stream_offset is None:
            stream_offset = ((64 + 8 * (nblocks + 1) + 15) // 16 * 16) % (1 << 64)
        h = np.zeros(1, oroch_amd.capi.HEADER_DTYPE)
        h["magic"], h["dtype"], h["block_values"] = b"OROCHBK1", dtype, 256
        h["nvalues"], h["nblocks"] = nvalues, nblocks
        h["table_offset"], h["stream_offset"], h["stream_bytes"] = 64, stream_offset, stream_bytes
        return h

    def check(h, nbytes):
        return L.oroch_container_check(h.ctypes.data, nbytes)

    assert check(header(0), 80) == 0
    assert check(header(1000, stream_bytes=160), 64 + 40 + 8 + 160) == 0
    # the wrap: (nvalues + 255) / 256 == 0 for these
    for nvalues in ((1 << 64) - 1, (1 << 64) - 255, (1 << 64) - 200):
        assert check(header(nvalues, nblocks=0), 80) != 0
        assert check(header(nvalues, nblocks=0, stream_offset=80), 1 << 20) != 0
        assert check(header(nvalues), 1 << 20) != 0
    # more blocks than the bytes could hold a table for
    assert check(header(1 << 40), 1 << 20) != 0
    assert check(header(256 * 1000), 4000) != 0
    # stream beyond the end, misaligned stream, wrong block count, bad dtype
    assert check(header(1000, stream_bytes=176), 64 + 40 + 8 + 160) != 0
    assert check(header(1000, stream_bytes=152), 1 << 20) != 0
    assert check(header(1000, nblocks=5), 1 << 20) != 0
    assert check(header(1000, dtype=9), 1 << 20) != 0
    # an array refuses such a file without touching the table (no GPU needed to refuse)
    bad = np.zeros(80, np.uint8)
    bad[:64] = header((1 << 64) - 1, nblocks=0).view(np.uint8)
    assert L.oroch_container_decode_host(bad.ctypes.data, bad.size, 0, 0, None) != 0

"""Generate tests/golden/golden_v1.npz from the REAL reference.

Run in a container that has /root/reference and oracle/_ref built
(`make -C oracle`):   python tests/golden/make_golden.py

Every expected output in the fixture is produced by the reference's own
headers (through oracle/ref_shim.cc); the fixture holds data only: inputs,
parameters and the bytes / fields the reference produced.  It travels with the
repository so that the oracle and the HIP path can be checked where the
reference tree does not exist.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import datagen as dg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

META_FIELDS = [f for f, _ in po.Meta._fields_]


def main():
    r = po.ref
    assert r is not None, "build oracle/_ref first (make -C oracle)"
    arrays = {}
    cases = []

    def add(kind, a, **kw):
        i = len(cases)
        a = np.ascontiguousarray(a)
        arrays["c%d_in" % i] = a.view(np.uint8)
        c = {"kind": kind, "dtype": po.dtype_code(a.dtype), "n": int(a.size)}
        c.update(kw)
        cases.append(c)
        return i

    rng = np.random.default_rng(20251111)

    # ---- per-block selection + integer_group image, all types, edge lengths
    lengths = [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 31, 63, 64, 65, 127, 128,
               129, 255, 256]
    for dt in dg.DTYPES:
        for rep in range(3):
            for n in lengths:
                a = dg.fuzz_block(rng, dt, n)
                i = add("group", a)
                m = r.select(a)
                cases[i]["meta"] = [int(x) for x in m.astuple()]
                arrays["c%d_out" % i] = r.group_encode(a)
        # 256-value blocks of every fuzz family, several draws
        for rep in range(40):
            a = dg.fuzz_block(rng, dt, 256)
            i = add("group", a)
            cases[i]["meta"] = [int(x) for x in r.select(a).astuple()]
            arrays["c%d_out" % i] = r.group_encode(a)

    # the headline distributions, a few blocks each, as multi-block streams
    for dt, kw in [(np.uint32, dict(seed=3)), (np.int32, dict(seed=3, shift=-200)),
                   (np.int64, dict(seed=4, body_bits=12, out_bits=40)),
                   (np.int16, dict(seed=5, body_bits=9, out_bits=16, permille=5, shift=-256)),
                   (np.uint64, dict(seed=6, body_bits=20, out_bits=50, shift=1 << 40))]:
        a = dg.outlier_mix(256 * 12 + 77, dt, **kw)
        i = add("blocks", a, block=256, align=16)
        out, offs = r.blocks_encode(a, 256, 16)
        arrays["c%d_out" % i] = out
        arrays["c%d_offs" % i] = offs

    # ---- whole-sequence bitpck: every width, partial last block
    for dt in dg.DTYPES:
        W = np.dtype(dt).itemsize * 8
        for nbits in range(1, W + 1):
            cap = 128 // nbits
            for n in sorted({1, cap - 1, cap, cap + 1, 3 * cap + 2} - {0}):
                if nbits not in (1, 2, 3, 7, 8, 13, 31, 32, 33, 42, 43, 63, 64, W - 1, W) and \
                        n != cap + 1:
                    continue
                a = dg.uniform_bits(n, dt, 100 + nbits, min(W, nbits + (nbits % 3 == 0)))
                i = add("bitpck", a, nbits=nbits)
                arrays["c%d_out" % i] = r.bitpck_encode(a, nbits)

    # ---- bitfor with extreme origins
    for dt in dg.DTYPES:
        W = np.dtype(dt).itemsize * 8
        info = np.iinfo(dt)
        for origin in (int(info.min), 0, int(info.max) - 1000 if W > 16 else int(info.max) - 20):
            for nbits in (1, 5, W // 2, W - 1, W):
                a = dg.uniform_bits(77, dt, 7 + nbits, min(nbits, W), shift=origin)
                i = add("bitfor", a, nbits=nbits, origin=po.raw64(origin, dt))
                arrays["c%d_out" % i] = r.bitfor_encode(a, origin, nbits)

    # ---- stand-alone bitpfr (packed low bits + exception lists)
    for dt in dg.DTYPES:
        W = np.dtype(dt).itemsize * 8
        for nbits in (1, 3, W // 2, W - 1):
            for n in (1, 9, 100, 1000):
                a = dg.outlier_mix(n, dt, 50 + nbits, body_bits=nbits, out_bits=W,
                                   permille=60, shift=-37)
                origin = int(a.min())
                i = add("bitpfr", a, nbits=nbits, origin=po.raw64(origin, dt))
                out, idx, val = r.bitpfr_encode(a, origin, nbits)
                arrays["c%d_out" % i] = out
                arrays["c%d_idx" % i] = idx
                arrays["c%d_val" % i] = val

    # ---- varint sequences
    for dt in dg.DTYPES:
        a = dg.fuzz_block(rng, dt, 200)
        i = add("varint", a)
        arrays["c%d_out" % i] = r.varint_encode(a)

    arrays["cases"] = np.frombuffer(json.dumps({"meta_fields": META_FIELDS, "cases": cases})
                                    .encode(), np.uint8)
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, len(cases), "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

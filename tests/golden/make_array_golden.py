"""Generate tests/golden/array_v1.npz from the REAL reference's integer_array.

Run where /root/reference exists and oracle/_ref is built (`make -C oracle`):
    python tests/golden/make_array_golden.py

For every element type: a few thousand values go into the reference's
integer_array (insert at the end, one by one), some insertions at other
positions follow -- the reference accepts only those whose position within its
group does not exceed the length of the tail (integer_array.h:221) and the
fixture records which ones it took -- and then a set of values is looked up
with find().  The fixture holds data only: the inputs, the positions found and
the text of group_info().
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

DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]


def main():
    assert po.ref is not None, "build oracle/_ref first (make -C oracle)"
    arrays, cases = {}, []
    for code, dt in enumerate(DTYPES):
        dt = np.dtype(dt)
        for variant, (nblocks, last) in enumerate(((9, 40), (4, 0), (0, 17))):
            rng = np.random.default_rng(20260 + 10 * code + variant)
            parts = [dg.fuzz_block(rng, dt, 256) for _ in range(nblocks)]
            if last:
                parts.append(dg.fuzz_block(rng, dt, last))
            values = np.concatenate(parts)
            n = values.size
            # positions the reference takes (small index within the group) and some it refuses
            ins_pos = np.array([0, n, 256 + min(last, 3) if n > 300 else 1, 700 if n > 800 else 2,
                                3 * 256 if n > 800 else 0, 100, 0], np.uint64)
            ins_val = dg.fuzz_block(rng, dt, ins_pos.size)
            there = values[rng.integers(0, n, size=12)]
            needles = np.concatenate([there, ins_val, dg.fuzz_block(rng, dt, 8),
                                      np.array([0, 1, np.iinfo(dt).min, np.iinfo(dt).max], dt)])
            found, info, taken = po.ref_array_script(values, ins_pos, ins_val, needles)
            i = len(cases)
            arrays["a%d_values" % i] = values.view(np.uint8)
            arrays["a%d_ins_pos" % i] = ins_pos
            arrays["a%d_ins_val" % i] = ins_val.view(np.uint8)
            arrays["a%d_taken" % i] = taken.astype(np.uint8)
            arrays["a%d_needles" % i] = needles.view(np.uint8)
            arrays["a%d_found" % i] = found
            arrays["a%d_info" % i] = np.frombuffer(info.encode("latin-1"), np.uint8)
            cases.append({"dtype": code, "n": int(n)})
    arrays["cases"] = np.frombuffer(json.dumps({"cases": cases}).encode(), np.uint8)
    out = os.path.join(HERE, "array_v1.npz")
    np.savez_compressed(out, **arrays)
    print("wrote", out, len(cases), "cases", os.path.getsize(out), "bytes;",
          "insertions taken:", int(sum(arrays["a%d_taken" % i].sum() for i in range(len(cases)))),
          "refused:", int(sum((1 - arrays["a%d_taken" % i]).sum() for i in range(len(cases)))))


if __name__ == "__main__":
    main()

"""Prints the blocks whose descriptor differs from the oracle's selection on fuzz data."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oroch_amd, datagen as dg
from oracle import pyoracle as po
dt = np.dtype(sys.argv[1] if len(sys.argv) > 1 else "uint32")
rng = np.random.default_rng(5)
a = np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(400)])
d = oroch_amd.group_select(torch.from_numpy(a).cuda())
bad = 0
for b in range(400):
    blk = a[b * 256:(b + 1) * 256]
    m = po.oracle.select(blk)
    row = d[b]
    got = tuple(int(row[k]) for k in ("encoding", "nbits", "dataspace", "metaspace", "noutliers", "index_encoding", "index_nbits", "value_encoding", "value_nbits"))
    want = (m.encoding, m.nbits, m.dataspace, m.metaspace, m.noutliers, m.index_encoding, m.index_nbits, m.value_encoding, m.value_nbits)
    if got[:5] != want[:5] and bad < 8:
        print(b, "got", got, "want", want)
        bad += 1
print("bad", bad)

"""The whole-sequence varint codec on the bench distribution (int32, 2^28 values), a few calls
each way; run under `rocprofv3 --kernel-trace --stats` for the kernels' own times, as
tools/prof_varint.sh does."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oroch_amd, devgen
dev = torch.device("cuda", 0)
n = 1 << 28
x = devgen.outlier_mix(n, torch.int64, 3, dev, shift=-200).to(torch.int32)
v = oroch_amd.varint_encode(x)
for _ in range(5):
    v = oroch_amd.varint_encode(x)
    y = oroch_amd.varint_decode(v, n, np.int32)
torch.cuda.synchronize()
print("bytes", v.numel(), "round trip", "ok" if torch.equal(x, y) else "MISMATCH")

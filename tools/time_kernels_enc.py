"""Per-kernel times of the separate-launch encode path only (no decode, so experiment
builds that leave parts of the stream unwritten are safe): OROCH_HIP_LIBRARY=... LOG2N=28"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oroch_amd
from bench import gen_outlier_mix_torch
log2n = int(os.environ.get("LOG2N", "28"))
n = 1 << log2n
x = gen_outlier_mix_torch(n, 3, "cuda")
enc = oroch_amd.GroupEncoder(np.int32, n, x.device)
oroch_amd.set_option("encode_path", 1)
for it in range(3): enc.encode(x)
torch.cuda.synchronize()
oroch_amd.profile_start(256)
for it in range(10): enc.encode(x)
torch.cuda.synchronize()
p = oroch_amd.profile_stop()
print(os.path.basename(os.environ.get("OROCH_HIP_LIBRARY", "default")), "ms at 2^%d:" % log2n, json.dumps({k: round(ms / c, 4) for k, (ms, c) in p.items()}))

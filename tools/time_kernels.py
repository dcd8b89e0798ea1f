"""Times the group kernels of whatever library OROCH_HIP_LIBRARY names (two-pass and single-pass)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oroch_amd
from bench import gen_outlier_mix_torch
log2n = int(os.environ.get("LOG2N", "28"))
n = 1 << log2n
x = gen_outlier_mix_torch(n, 3, "cuda")
enc = oroch_amd.GroupEncoder(np.int32, n, x.device)
back = torch.empty_like(x)
res = {}
for mode in ("two", "one"):
    oroch_amd.set_option("encode_path", 1 if mode == "two" else 2)
    for it in range(2): enc.encode(x)
    torch.cuda.synchronize()
    oroch_amd.profile_start(256)
    for it in range(5):
        enc.encode(x)
        oroch_amd.lib.oroch_group_decode(4, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(), oroch_amd.capi._stream())
    torch.cuda.synchronize()
    p = oroch_amd.profile_stop()
    res[mode] = {k: round(ms / c * (1 << 30) / n, 3) for k, (ms, c) in p.items()}
print(os.environ.get("OROCH_HIP_LIBRARY", "default"), "ms scaled to 2^30 values:", json.dumps(res), "ok" if torch.equal(back, x) else "MISMATCH")

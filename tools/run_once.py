"""One encode (two-pass and single-pass) + decode at 2^LOG2N values, for counter collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oroch_amd
from bench import gen_outlier_mix_torch
n = 1 << int(os.environ.get("LOG2N", "26"))
x = gen_outlier_mix_torch(n, 3, "cuda")
enc = oroch_amd.GroupEncoder(np.int32, n, x.device)
back = torch.empty_like(x)
for mode in ("two", "one"):
    oroch_amd.set_option("encode_path", 1 if mode == "two" else 2)
    for it in range(2):
        enc.encode(x)
        oroch_amd.lib.oroch_group_decode(4, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(), oroch_amd.capi._stream())
torch.cuda.synchronize()
print("ok", torch.equal(back, x))

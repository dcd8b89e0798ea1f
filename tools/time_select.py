"""Times the selection kernel alone (descriptors only, nothing downstream reads them, so
experiment builds whose descriptors are wrong are safe here):
OROCH_HIP_LIBRARY=tools/lib_x.so LOG2N=28 python3 tools/time_select.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oroch_amd
from bench import gen_outlier_mix_torch
log2n = int(os.environ.get("LOG2N", "28"))
n = 1 << log2n
x = gen_outlier_mix_torch(n, 3, "cuda")
desc = torch.empty(((n + 255) // 256) * 32, dtype=torch.uint8, device=x.device)
code = oroch_amd.dtype_code(np.int32)
def run():
    rc = oroch_amd.lib.oroch_group_select(code, x.data_ptr(), n, desc.data_ptr(), oroch_amd.capi._stream())
    assert rc == 0, rc
for it in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(10): run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(os.path.basename(os.environ.get("OROCH_HIP_LIBRARY", "default")), "select 2^%d: %.4f ms = %.3f of 8 TB/s reading" % (log2n, ms, 4 * n / (ms * 1e-3) / 8e12))

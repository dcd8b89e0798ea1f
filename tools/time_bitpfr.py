"""Times whole-sequence bitpfr alone (uint32, 10-bit body + 1 % outliers, 2^LOG2N values); run under
rocprofv3 --kernel-trace --stats for the kernels one by one."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
dev = torch.device("cuda", 0)
n = 1 << int(os.environ.get("LOG2N", "28"))
x = devgen.outlier_mix(n, torch.int64, 3, dev).to(torch.int32).view(torch.uint32)
b, idx, val, k = oroch_amd.bitpfr_encode(x, 0, 10, capacity=n // 64)
def run(name, fn, nbytes):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    oroch_amd.profile_start(256)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    p = oroch_amd.profile_stop()
    ms = {kk: round(v / c, 4) for kk, (v, c) in p.items()}
    tot = sum(ms.values())
    print(name, json.dumps(ms), "%.1f%% of 8 TB/s" % (nbytes / tot / 1e6 / 80))
run("encode", lambda: oroch_amd.bitpfr_encode(x, 0, 10, capacity=n // 64), 4 * n + b.numel() + 16 * k)
run("decode", lambda: oroch_amd.bitpfr_decode(b, n, 0, 10, idx, val, np.uint32), 4 * n + b.numel() + 16 * k)

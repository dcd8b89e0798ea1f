"""Times the group kernels on each of the eight block shapes of the int64 mix alone (2^LOG2N values of one
shape): which shapes the mix's time comes from.  DT=int64|int32"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
log2n = int(os.environ.get("LOG2N", "26"))
dt = getattr(torch, os.environ.get("DT", "int64"))
n = 1 << log2n
dev = torch.device("cuda", 0)
names = ["constant", "12-bit", "12-bit+1% 40-bit", "signed geometric", "far origin", "raw", "ramp", "3-bit"]
mix = devgen.block_mix_int64(8 * n, 4, dev)
code = oroch_amd.dtype_code(dt)
for k, name in enumerate(names):
    x = mix.view(-1, 8, 256)[:, k, :].contiguous().view(-1).to(dt)
    enc = oroch_amd.GroupEncoder(x.dtype, n, x.device)
    back = torch.empty_like(x)
    for it in range(2): enc.encode(x)
    torch.cuda.synchronize()
    oroch_amd.profile_start(256)
    for it in range(5):
        enc.encode(x)
        oroch_amd.lib.oroch_group_decode(code, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(), oroch_amd.capi._stream())
    torch.cuda.synchronize()
    p = oroch_amd.profile_stop()
    total = int(enc.offsets[-1].item())
    raw = n * x.element_size()
    res = {kk: round(ms / c, 4) for kk, (ms, c) in p.items()}
    fr = {kk: round((raw + (total if kk != "select" else 0)) / (v * 1e-3) / 8e12, 3) for kk, v in res.items() if kk != "scan"}
    encs = np.bincount(oroch_amd.group_select(x[:256 * 64])["encoding"], minlength=7).tolist()
    print("%-18s ratio %.3f encodings %s ms %s frac %s %s" % (name, total / raw, encs, json.dumps(res), json.dumps(fr), "ok" if torch.equal(back, x) else "MISMATCH"))

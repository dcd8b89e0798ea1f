"""Times the group kernels for a dtype/distribution: DT=int64 LOG2N=27 python3 tools/time_any.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
dt = os.environ.get("DT", "int64")
log2n = int(os.environ.get("LOG2N", "27"))
kind = os.environ.get("KIND", "outlier")
n = 1 << log2n
dev = torch.device("cuda", 0)
unsigned64 = dt == "uint64"  # generated as int64 bit patterns, handed over as uint64
tdt = getattr(torch, "int64" if unsigned64 else dt)
if kind == "outlier":
    x = devgen.outlier_mix(n, tdt, int(os.environ.get("SEED", "3")), dev, body_bits=int(os.environ.get("BODY", "10")), out_bits=int(os.environ.get("OUT", "28")), shift=int(os.environ.get("SHIFT", "-200")),
                           **({"permille": int(os.environ["PERMILLE"])} if "PERMILLE" in os.environ else {}))
elif kind == "mix":
    x = devgen.block_mix_int64(n, 4, dev).to(tdt)
else:
    x = devgen.uniform_bits(n, tdt, 3, int(os.environ.get("BODY", "13")), dev)
if unsigned64:
    x = (x + (1 << 40)).view(torch.uint64)
enc = oroch_amd.GroupEncoder(x.dtype, n, x.device)
back = torch.empty_like(x)
code = oroch_amd.dtype_code(x.dtype)
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
res = {k: round(ms / c, 4) for k, (ms, c) in p.items()}
fr = {k: round((raw + (total if k != "select" else 0)) / (v * 1e-3) / 8e12, 3) for k, v in res.items() if k != "scan"}
print(dt, kind, "n=2^%d" % log2n, "ratio %.3f" % (total / raw), "ms:", json.dumps(res), "frac of 8TB/s:", json.dumps(fr), "ok" if torch.equal(back.view(torch.uint8), x.view(torch.uint8)) else "MISMATCH")

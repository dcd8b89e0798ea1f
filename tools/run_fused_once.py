"""One or two encodes by one path at 2^LOG2N int32 values, for counter collection:
ENCODE_PATH=3 LOG2N=26 python3 tools/run_fused_once.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
n = 1 << int(os.environ.get("LOG2N", "26"))
kind = os.environ.get("KIND", "outlier")
dev = torch.device("cuda", 0)
if kind == "outlier":
    x = devgen.outlier_mix(n, torch.int32, 3, dev, body_bits=10, out_bits=28, shift=-200)
else:
    x = devgen.uniform_bits(n, torch.int32, 3, int(os.environ.get("BODY", "13")), dev)
enc = oroch_amd.GroupEncoder(np.int32, n, dev)
oroch_amd.set_option("encode_path", int(os.environ.get("ENCODE_PATH", "3")))
for it in range(int(os.environ.get("REPS", "2"))):
    enc.encode(x)
torch.cuda.synchronize()
if os.environ.get("DECODE"):
    back = torch.empty_like(x)
    oroch_amd.lib.oroch_group_decode(4, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(), oroch_amd.capi._stream())
    torch.cuda.synchronize()
    print("roundtrip", torch.equal(back, x))
print("done", int(enc.offsets[-1].item()))

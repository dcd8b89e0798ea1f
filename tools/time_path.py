"""Times encode by one path: ENCODE_PATH=3 LOG2N=28 REPS=10 python3 tools/time_path.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
n = 1 << int(os.environ.get("LOG2N", "28"))
dev = torch.device("cuda", 0)
x = devgen.outlier_mix(n, torch.int32, 3, dev, body_bits=10, out_bits=28, shift=-200)
enc = oroch_amd.GroupEncoder(np.int32, n, dev)
oroch_amd.set_option("encode_path", int(os.environ.get("ENCODE_PATH", "3")))
reps = int(os.environ.get("REPS", "10"))
for it in range(reps):
    torch.cuda.synchronize()
    t0 = time.time()
    enc.encode(x)
    torch.cuda.synchronize()
    print("call %d: %.3f ms" % (it, (time.time() - t0) * 1e3), flush=True)
print("total", int(enc.offsets[-1].item()))
torch.cuda.synchronize()
t0 = time.time()
for it in range(reps):
    enc.encode(x)
torch.cuda.synchronize()
print("back to back: %.3f ms per call" % ((time.time() - t0) * 1e3 / reps), flush=True)
back = torch.empty_like(x)
oroch_amd.lib.oroch_group_decode(4, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(), oroch_amd.capi._stream())
torch.cuda.synchronize()
print("roundtrip", bool(torch.equal(back, x)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(reps):
    enc.encode(x)
e1.record()
torch.cuda.synchronize()
print("back to back by events: %.3f ms per call" % (e0.elapsed_time(e1) / reps), flush=True)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
n = 1 << int(os.environ.get("LOG2N", "28"))
dev = torch.device("cuda", 0)
x = devgen.outlier_mix(n, torch.int32, 3, dev, body_bits=10, out_bits=28, shift=-200)
enc = oroch_amd.GroupEncoder(np.int32, n, dev)
for path in (1, 3, 1, 3):
    oroch_amd.set_option("encode_path", path)
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        enc.encode(x)
        torch.cuda.synchronize()
        print("path %d call %d: %.3f ms" % (path, it, (time.time() - t0) * 1e3), flush=True)

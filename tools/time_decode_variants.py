"""Encodes with the default library, then times oroch_group_decode of every library given on the command line (experiment builds from tools/build_variants.sh)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oroch_amd
from bench import gen_outlier_mix_torch
n = 1 << int(os.environ.get("LOG2N", "26"))
x = gen_outlier_mix_torch(n, 3, "cuda")
if os.environ.get("KIND") == "uniform":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import devgen
    x = devgen.uniform_bits(n, torch.int32, 3, int(os.environ.get("BODY", "13")), torch.device("cuda", 0))
enc = oroch_amd.GroupEncoder(np.int32, n, x.device)
enc.encode(x)
back = torch.empty_like(x)
torch.cuda.synchronize()
for path in sys.argv[1:]:
    L = ctypes.CDLL(path)
    L.oroch_group_decode.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        if it == 1: ev[0].record()
        L.oroch_group_decode(4, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(), None)
    ev[1].record()
    torch.cuda.synchronize()
    print(os.path.basename(path), "decode ms scaled to 2^30: %.3f" % (ev[0].elapsed_time(ev[1]) / 2 * (1 << 30) / n), "ok" if torch.equal(back, x) else "MISMATCH")

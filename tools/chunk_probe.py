"""Does cutting the encode into pieces that fit the last-level cache pay?  The
two passes over a piece then read its values from memory once and from the
cache once.  python3 tools/chunk_probe.py [log2n]"""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oroch_amd, devgen
from oroch_amd import capi
dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << L
x = devgen.outlier_mix(n, torch.int32, 3, dev, body_bits=10, out_bits=28, shift=-200)
code = oroch_amd.dtype_code(x.dtype)
lib = oroch_amd.lib
for lc in (L, 27, 26, 25, 24, 23):
    if lc > L:
        continue
    c = 1 << lc
    pieces = n // c
    bound = lib.oroch_group_bound(code, c)
    wsb = lib.oroch_group_workspace(code, c)
    out = torch.empty(bound * pieces, dtype=torch.uint8, device=dev)
    offs = torch.empty((c // 256 + 1) * pieces, dtype=torch.int64, device=dev)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    def run():
        for i in range(pieces):
            rc = lib.oroch_group_encode(code, x.data_ptr() + 4 * c * i, c, out.data_ptr() + bound * i, bound,
                                        offs.data_ptr() + 8 * (c // 256 + 1) * i, ws.data_ptr(), wsb, None, capi._stream())
            assert rc == 0
    run(); run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for it in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    oroch_amd.profile_start(4096)
    run()
    torch.cuda.synchronize()
    p = oroch_amd.profile_stop()
    print("pieces of 2^%d: %.3f ms per 2^%d values; kernels %s" % (lc, ms, L, json.dumps({k: round(v[0], 3) for k, v in p.items()})), flush=True)
    del out, offs, ws

"""Kernel times of the block path on the inputs the reviews name, one line each:
python3 tools/time_inputs.py [headline int16cfg5 uniform13 int16 ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oroch_amd, devgen
dev = torch.device("cuda", 0)
CASES = {
    "headline": lambda: devgen.outlier_mix(1 << 28, torch.int32, 3, dev, body_bits=10, out_bits=28, shift=-200),
    "int16cfg5": lambda: devgen.outlier_mix(1 << 28, torch.int16, 5, dev, body_bits=9, out_bits=16, permille=5, shift=-256),
    "uniform13": lambda: devgen.uniform_bits(1 << 28, torch.int32, 3, 13, dev),
    "int16": lambda: devgen.outlier_mix(1 << 29, torch.int16, 3, dev, body_bits=9, out_bits=15, shift=-200),
    "uint64": lambda: (devgen.outlier_mix(1 << 27, torch.int64, 6, dev, body_bits=20, out_bits=50, shift=1 << 40)).view(torch.uint64),
    "uint64b": lambda: (devgen.outlier_mix(1 << 28, torch.int64, 3, dev, body_bits=20, out_bits=50, shift=-200) + (1 << 40)).view(torch.uint64),
    "int64mix": lambda: devgen.block_mix_int64(1 << 28, 4, dev),
    "raw32": lambda: devgen.uniform_bits(1 << 28, torch.int32, 3, 32, dev),
    "const32": lambda: (torch.arange(1 << 28, dtype=torch.int64, device=dev) >> 8).to(torch.int32),
    "int8": lambda: devgen.outlier_mix(1 << 29, torch.int8, 3, dev, body_bits=4, out_bits=7, shift=-5),
}
for name in (sys.argv[1:] or list(CASES)):
    x = CASES[name]()
    n = x.numel()
    enc = oroch_amd.GroupEncoder(x.dtype, n, dev)
    back = torch.empty_like(x)
    code = oroch_amd.dtype_code(x.dtype)
    for it in range(2):
        enc.encode(x)
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
    encp = sum(v for k, v in res.items() if k != "decode")
    fr = {"encode path": round((raw + total) / (encp * 1e-3) / 8e12, 3), "decode": round((raw + total) / (res["decode"] * 1e-3) / 8e12, 3)}
    ok = torch.equal(back.view(torch.uint8), x.view(torch.uint8))
    print("%-10s n=2^%d ratio %.3f ms %s frac of 8TB/s %s %s" % (name, n.bit_length() - 1, total / raw, json.dumps(res), json.dumps(fr), "ok" if ok else "MISMATCH"), flush=True)
    del x, enc, back

"""Times the whole-sequence codecs (bitpck, bitfor, bitpfr with caller-supplied parameters) on the
benchmark configurations: uint32 13-bit, int32 16-bit range with an origin, uint32 with 1 % outliers."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
dev = torch.device("cuda", 0)

def run(name, fn, raw_bytes, packed_bytes):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    oroch_amd.profile_start(256)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    p = oroch_amd.profile_stop()
    ms = {k: round(v / c, 4) for k, (v, c) in p.items()}
    tot = sum(ms.values())
    print(name, "ms", json.dumps(ms), "-> %.0f GB/s of raw+packed, %.1f%% of 8 TB/s" % ((raw_bytes + packed_bytes) / tot / 1e6, (raw_bytes + packed_bytes) / tot / 1e6 / 80))

n = 1 << 28
x = devgen.uniform_bits(n, torch.int32, 1, 13, dev).view(torch.uint32)
b = oroch_amd.bitpck_encode(x, 13)
run("bitpck uint32 13-bit encode 2^28", lambda: oroch_amd.bitpck_encode(x, 13), 4 * n, b.numel())
run("bitpck uint32 13-bit decode 2^28", lambda: oroch_amd.bitpck_decode(b, n, 13, np.uint32), 4 * n, b.numel())
del x, b
n = 1 << 26
x = devgen.uniform_bits(n, torch.int32, 2, 16, dev, shift=980000)
lo = int(x.min())
f = oroch_amd.bitfor_encode(x, lo, 16)
run("bitfor int32 16-bit encode 2^26", lambda: oroch_amd.bitfor_encode(x, lo, 16), 4 * n, f.numel())
run("bitfor int32 16-bit decode 2^26", lambda: oroch_amd.bitfor_decode(f, n, lo, 16, np.int32), 4 * n, f.numel())
del x, f
n = 1 << 28
x = devgen.outlier_mix(n, torch.int64, 3, dev).to(torch.int32).view(torch.uint32)
b, idx, val, k = oroch_amd.bitpfr_encode(x, 0, 10, capacity=n // 64)
run("bitpfr uint32 10-bit + 1%% encode 2^28 (%d exceptions)" % k, lambda: oroch_amd.bitpfr_encode(x, 0, 10, capacity=n // 64), 4 * n, b.numel() + 16 * k)
run("bitpfr uint32 10-bit + 1% decode 2^28", lambda: oroch_amd.bitpfr_decode(b, n, 0, 10, idx, val, np.uint32), 4 * n, b.numel() + 16 * k)
n = 1 << 28
x = devgen.outlier_mix(n, torch.int64, 3, dev, shift=-200).to(torch.int32)
v = oroch_amd.varint_encode(x)
run("varint int32 (bench distribution) encode 2^28", lambda: oroch_amd.varint_encode(x), 4 * n, v.numel())
run("varint int32 (bench distribution) decode 2^28", lambda: oroch_amd.varint_decode(v, n, np.int32), 4 * n, v.numel())

"""Compares the single-launch encoder (encode path 3) with the separate-launch path (1) on int32
shapes, then times both: OROCH_HIP_LIBRARY=tools/lib_dev.so LOG2N=28 python3 tools/check_fused.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oroch_amd, devgen
import datagen as dg

PATH = int(os.environ.get("FUSED_PATH", "3"))
dev = torch.device("cuda", 0)


def encode(x, path):
    with oroch_amd.option("encode_path", path):
        s, o = oroch_amd.group_encode(x)
    torch.cuda.synchronize()
    return s, o


bad = 0
rng = np.random.default_rng(1)
cases = []
if not os.environ.get("SKIP_CHECK"):
    for n in (1, 255, 256, 257, 1024, 1025, 256 * 64 + 100, 256 * 4 * 64 * 3 + 5, 256 * 4 * 64 * 70 + 256 * 3):
        cases.append(("outlier n=%d" % n, dg.outlier_mix(n, np.int32, seed=3, shift=-200)))
    cases.append(("fuzz", np.concatenate([dg.fuzz_block(rng, np.int32, 256) for _ in range(3000)] + [dg.fuzz_block(rng, np.int32, 77)])))
    cases.append(("uniform13", dg.uniform_bits(256 * 1000, np.int32, 5, 13)))
    cases.append(("big", dg.outlier_mix(1 << 24, np.int32, seed=12, shift=-200)))
    for name, a in cases:
        x = torch.from_numpy(a).to(dev)
        s1, o1 = encode(x, 1)
        s3, o3 = encode(x, PATH)
        ok = torch.equal(o1, o3) and torch.equal(s1, s3)
        back = oroch_amd.group_decode(s3, o3, a.size, np.int32)
        ok2 = torch.equal(back, x)
        print(name, "same" if ok else "DIFFERENT", "roundtrip ok" if ok2 else "ROUNDTRIP BAD", flush=True)
        if not ok:
            bad += 1
            d = (o1 != o3).nonzero()
            print("  first offset difference at", d[:4].flatten().tolist(), "total", int(o1[-1]), int(o3[-1]))
            if o1[-1] == o3[-1]:
                dd = (s1 != s3).nonzero()
                print("  first byte difference at", dd[:4].flatten().tolist())

log2n = int(os.environ.get("LOG2N", "28"))
n = 1 << log2n
kind = os.environ.get("KIND", "outlier")
if kind == "outlier":
    x = devgen.outlier_mix(n, torch.int32, 3, dev, body_bits=10, out_bits=28, shift=-200)
else:
    x = devgen.uniform_bits(n, torch.int32, 3, int(os.environ.get("BODY", "13")), dev)
enc = oroch_amd.GroupEncoder(np.int32, n, dev)
res = {}
for path in (1, PATH):
    oroch_amd.set_option("encode_path", path)
    for it in range(3):
        enc.encode(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for it in range(reps):
        enc.encode(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    total = int(enc.offsets[-1].item())
    res[path] = (round(ms, 4), round((4 * n + total) / (ms * 1e-3) / 8e12, 3))
oroch_amd.set_option("encode_path", 0)
print("n=2^%d" % log2n, kind, "ms, frac of 8 TB/s by path:", json.dumps(res), "FAILURES %d" % bad if bad else "all same")

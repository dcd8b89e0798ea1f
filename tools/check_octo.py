"""Compares the eight-blocks-per-wavefront kernels with the separate launches of four blocks per
wavefront on many shapes and element types (descriptors of selection, streams of the single
launch), then times both: LOG2N=28 python3 tools/check_octo.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oroch_amd, devgen
import datagen as dg

PATH = int(os.environ.get("FUSED_PATH", "4"))
dev = torch.device("cuda", 0)


def encode(x, path):
    with oroch_amd.option("encode_path", path):
        s, o = oroch_amd.group_encode(x)
    torch.cuda.synchronize()
    return s, o


def select(x, how):
    with oroch_amd.option("select_one_block", how):
        d = oroch_amd.group_select(x)
    torch.cuda.synchronize()
    return d


bad = 0
rng = np.random.default_rng(1)
if not os.environ.get("SKIP_CHECK"):
    for dt in [np.dtype(t).type for t in os.environ.get("DTYPES", "int32 uint32 int16 uint16 int8 uint8").split()]:
        cases = []
        for n in (1, 255, 2048, 2049, 256 * 64 + 100, 256 * 8 * 11 * 3 + 5, 256 * 8 * 11 * 70 + 256 * 3):
            cases.append(("outlier n=%d" % n, dg.outlier_mix(n, dt, seed=3, shift=-20 if np.dtype(dt).kind == "i" else 0,
                                                             body_bits=min(10, np.dtype(dt).itemsize * 8 - 3),
                                                             out_bits=np.dtype(dt).itemsize * 8 - 4)))
        cases.append(("fuzz", np.concatenate([dg.fuzz_block(rng, dt, 256) for _ in range(4000)] + [dg.fuzz_block(rng, dt, 77)])))
        cases.append(("uniform", dg.uniform_bits(256 * 1000, dt, 5, np.dtype(dt).itemsize * 8 - 3)))
        for name, a in cases:
            x = torch.from_numpy(a).to(dev)
            d0, d2 = select(x, 0), select(x, 2)
            oksel = bool(np.array_equal(np.asarray(d0), np.asarray(d2))) if not torch.is_tensor(d0) else torch.equal(d0, d2)
            s1, o1 = encode(x, 1)
            s3, o3 = encode(x, PATH)
            ok = torch.equal(o1, o3) and torch.equal(s1, s3)
            back = oroch_amd.group_decode(s3, o3, a.size, dt)
            ok2 = torch.equal(back, x)
            print(np.dtype(dt).name, name, "select same" if oksel else "SELECT DIFFERENT", "same" if ok else "DIFFERENT",
                  "roundtrip ok" if ok2 else "ROUNDTRIP BAD", flush=True)
            if not oksel:
                bad += 1
                a0, a2 = np.asarray(d0), np.asarray(d2)
                w = np.nonzero(a0 != a2)[0]
                print("  first descriptors that differ:", w[:4].tolist())
                for i in w[:2]:
                    print("   ", a0[i], "\n   ", a2[i])
            if not ok:
                bad += 1
                d = (o1 != o3).nonzero()
                print("  first offset difference at", d[:4].flatten().tolist(), "total", int(o1[-1]), int(o3[-1]))
                if o1[-1] == o3[-1]:
                    dd = (s1 != s3).nonzero()
                    print("  first byte difference at", dd[:4].flatten().tolist())
                    at = int(dd[0])
                    oo = o1.cpu().numpy()
                    blk = int(np.searchsorted(oo, at, side="right")) - 1
                    lo, hi = int(oo[blk]), int(oo[blk + 1])
                    print("  block", blk, "at", lo, "size", hi - lo, "descriptor", np.asarray(d0)[blk])
                    b1, b3 = s1[lo:hi].cpu().numpy(), s3[lo:hi].cpu().numpy()
                    w = np.nonzero(b1 != b3)[0]
                    print("  bytes that differ (relative):", w[:40].tolist(), "of", hi - lo)
                    print("  expected", b1[max(0, w[0] - 8):w[0] + 24].tolist())
                    print("  got     ", b3[max(0, w[0] - 8):w[0] + 24].tolist())
                    print("  values", a[blk * 256:blk * 256 + 256].tolist())

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
        enc.encode(x, want_total=True)
    torch.cuda.synchronize()
    reps = 10
    oroch_amd.profile_start(64 * reps)
    for it in range(reps):
        enc.encode(x, want_total=True)
    torch.cuda.synchronize()
    prof = oroch_amd.profile_stop()
    ms = sum(t / c for t, c in prof.values())
    total = int(enc.offsets[-1].item())
    res[path] = (round(ms, 4), round((4 * n + total) / (ms * 1e-3) / 8e12, 3), {k: round(t / c, 4) for k, (t, c) in prof.items()})
oroch_amd.set_option("encode_path", 0)
# selection alone (kernel time by HIP events on the launch stream)
sel = {}
desc = torch.zeros((n // 256) * 32, dtype=torch.uint8, device=dev)
for how in (0, 2):
    oroch_amd.set_option("select_one_block", how)
    for it in range(2):
        oroch_amd.lib.oroch_group_select(4, x.data_ptr(), n, desc.data_ptr(), oroch_amd.capi._stream())
    torch.cuda.synchronize()
    oroch_amd.profile_start(64)
    for it in range(5):
        oroch_amd.lib.oroch_group_select(4, x.data_ptr(), n, desc.data_ptr(), oroch_amd.capi._stream())
    torch.cuda.synchronize()
    prof = oroch_amd.profile_stop()
    sel[how] = {k: round(ms / cnt, 4) for k, (ms, cnt) in prof.items()}
oroch_amd.set_option("select_one_block", 0)
print("n=2^%d" % log2n, kind, "encode ms, frac of 8 TB/s by path:", json.dumps(res), "select kernel ms:", json.dumps(sel),
      "FAILURES %d" % bad if bad else "all same")

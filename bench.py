#!/usr/bin/env python3
"""Headline benchmark: device-resident per-block encode + decode of int32 values.

Workload (BASELINE.json north star): 2^30 int32 values per GPU, 10-bit noise
with 1 % 28-bit outliers shifted by -200, cut into blocks of 256 values; every
block goes through statistics -> codec selection -> offsets scan -> encode
(mostly patched frame-of-reference), then the stream is decoded back.  One
"step" is one encode pass plus one decode pass over the whole array with the
input already resident in HBM.  value = uncompressed bytes through encode plus
uncompressed bytes through decode, per second, summed over all GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n L]

For N > 1 launch with torch.distributed.run (one rank per GPU); each rank owns
a contiguous block range (weak scaling: 2^L values per GPU) and only per-shard
byte counts are exchanged, after the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def gen_outlier_mix_torch(n, seed, device, body_bits=10, out_bits=28, permille=10, shift=-200,
                          start=0, chunk=1 << 26):
    """tests/datagen.outlier_mix for int32, generated on the device.

    int64 arithmetic wraps like uint64; logical right shifts are emulated.
    """
    import torch

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def s64(c):
        return c - (1 << 64) if c >= (1 << 63) else c

    out = torch.empty(n, dtype=torch.int32, device=device)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        i = torch.arange(start + lo, start + lo + m, dtype=torch.int64, device=device)
        z = i + seed + s64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        r = z ^ lsr(z, 31)
        v = r & ((1 << body_bits) - 1)
        hi = lsr(r, 32)
        # unsigned (r >> 32) % 1000 on a value below 2^32 is a plain remainder
        sel = (hi % 1000) < permille
        big = (lsr(r, 12) | (1 << (out_bits - 1))) & ((1 << out_bits) - 1)
        v = torch.where(sel, v | big, v) + shift
        out[lo:lo + m] = v.to(torch.int32)
    return out


def cpu_info():
    """Model name of the host CPU, the machine's logical CPUs and the share of
    them this process may use."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    # the GPU box hands a one-GPU job 16 of the machine's CPUs and says so in
    # OMP_NUM_THREADS; the affinity mask there still shows the whole machine
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or affinity
    return model, os.cpu_count(), max(1, min(share, affinity))


def cpu_blocks(po, a, threads):
    """select + encode + decode of blocks of 256 by the reference's own code,
    one thread and `threads` threads over disjoint block ranges; best of 2."""
    def run(t):
        best = None
        for _ in range(2):
            r = po.ref.bench_blocks(a, 256, t)
            assert r[4] == 1.0, "reference round trip failed"
            if best is None or r[:3].sum() < best[:3].sum():
                best = r
        return best
    one, many = run(1), run(threads)
    gbs = lambda r: 2 * a.nbytes / r[:3].sum() / 1e9
    return {"value": gbs(many), "value_1thread": gbs(one),
            "phases_1thread_GBps": {"select": a.nbytes / one[0] / 1e9, "encode": a.nbytes / one[1] / 1e9,
                                    "decode": a.nbytes / one[2] / 1e9}}


def cpu_whole_arrays(po, dg):
    """The whole-sequence codecs with the caller's parameters (the reference's
    bitpck.h / bitfor.h / bitpfr.h on one thread), the distributions of the
    configurations at 2^24 values: GB/s of uncompressed bytes."""
    n = 1 << 24
    out = {}
    a = dg.uniform_bits(n, np.uint32, 1, 13)
    cases = [("bitpck uint32 13 bits", a, 0, 0, 13)]
    a = dg.uniform_bits(n, np.int32, 2, 16, shift=-20000 + 1000000)
    cases.append(("bitfor int32 16 bits", a, 1, int(a.min()), 16))
    a = dg.outlier_mix(n, np.uint32, 3)
    cases.append(("bitpfr uint32 10 bits + 1% outliers", a, 2, 0, 10))
    for name, a, kind, origin, nbits in cases:
        best = None
        for _ in range(2):
            r = po.ref.bench_whole(a, kind, origin, nbits)
            assert r[2] == 1.0, "reference round trip failed"
            if best is None or r[:2].sum() < best[:2].sum():
                best = r
        out[name] = {"encode_GBps": a.nbytes / best[0] / 1e9, "decode_GBps": a.nbytes / best[1] / 1e9}
    return out


def cpu_native_baseline(log2n, config):
    """The CPU baseline once more with the reference built at -O3 -march=native of
    the machine that built the checker, in a child process (a host without those
    instructions only loses this figure).  Called before this process touches
    the GPU: no child is started from a process that has initialised it."""
    native = os.path.join(ROOT, "oracle", "_ref", "liboroch_ref_native.so")
    if not os.path.exists(native):
        return None
    import subprocess

    arch = "unknown"
    try:
        arch = open(os.path.join(ROOT, "oracle", "_ref", "native_arch.txt")).read().strip()
    except OSError:
        pass
    env = dict(os.environ, OROCH_REF_LIBRARY=os.path.join("_ref", "liboroch_ref_native.so"))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-native-child",
                        "--cpu-log2n", str(min(log2n, 26)), "--config", config],
                       env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode == 0 and lines:
        return dict(json.loads(lines[-1]), built_for=arch, sample="2^%d values, otherwise as above" % min(log2n, 26))
    return {"value": None, "built_for": arch,
            "note": "did not run on this host (exit %d): built for another CPU" % r.returncode}


def cpu_baseline(log2n, seed, config="headline", native_child=False, native=None):
    """The reference's own code (oracle/_ref) -- or the C port -- on host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datagen as dg
    from oracle import pyoracle as po

    n = 1 << log2n
    if config == "int64mix":
        a = dg.block_mix_int64(n, seed + 1)
    else:
        a = dg.outlier_mix(n, np.int32, seed=seed, shift=-200)
    name = str(a.dtype)
    nbytes = a.nbytes
    model, nproc, threads = cpu_info()
    if po.ref is not None:
        res = cpu_blocks(po, a, threads)
        if native_child:
            return res
        out = {
            "value": res["value"], "unit": "GB/s", "cores": threads, "kind": "reference",
            "cpu_model": model, "nproc": nproc,
            "value_1thread": res["value_1thread"], "phases_1thread_GBps": res["phases_1thread_GBps"],
            "whole_array_1thread_GBps": cpu_whole_arrays(po, dg),
            "sample": "2^%d %s values of the bench distribution, blocks of 256, select+encode+"
                      "decode by the reference headers (g++ -O2), best of 2, %d std::threads over "
                      "disjoint block ranges: every CPU this job may use (the machine has %d, a "
                      "one-GPU job's share is what OMP_NUM_THREADS says); whole_array: the "
                      "reference's bitpck / bitfor / bitpfr with the caller's parameters on 2^24 "
                      "values, one thread" % (log2n, name, threads, nproc),
        }
        if native is not None:
            out["O3_march_native"] = native
        return out
    o = po.oracle
    t0 = time.perf_counter()
    stream, offs = o.blocks_encode(a, 256, 16)
    t1 = time.perf_counter()
    back = o.blocks_decode(stream, offs, n, a.dtype, 256)
    t2 = time.perf_counter()
    assert np.array_equal(back, a)
    return {"value": 2 * nbytes / (t2 - t0) / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "cpu_model": model, "nproc": nproc,
            "sample": "2^%d %s values of the bench distribution through the C restatement "
                      "(encode %.2fs, decode %.2fs)" % (log2n, name, t1 - t0, t2 - t1)}


def measured_traffic(direction, log2n):
    """HBM bytes per step of one direction ("encode": every kernel of the
    encode path, "decode": the decode kernel) from the committed PMC passes
    (rocprofv3 cannot run inside this process), only for the workload they were
    collected on and only while the kernel sources are still the ones they were
    measured on: returns (bytes or None, what the record says about itself)."""
    import oroch_amd.sources as sources

    path = os.path.join(ROOT, "profiles", "traffic_int32_2p30.json")
    if log2n != 30 or not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    k = rec.get("kernels", {})
    about = {"commit": rec.get("commit"), "kernel_sources": rec.get("kernel_sources"),
             "kernel_sources_now": sources.digest()}
    if not k or about["kernel_sources"] != about["kernel_sources_now"]:
        about["stale"] = True
        return None, about
    decode = sum(v["total_bytes"] for name, v in k.items() if "decode" in name)
    encode = sum(v["total_bytes"] for name, v in k.items() if "decode" not in name)
    return (decode if direction == "decode" else encode), about


def end_to_end(x_dev, n):
    """Host buffer -> pinned staging -> device encode -> host stream, and back."""
    import ctypes as C

    import oroch_amd

    a = x_dev[:n].cpu().numpy()
    code = oroch_amd.dtype_code(a.dtype)
    lib = oroch_amd.lib
    bound = lib.oroch_group_bound(code, n)
    out = np.zeros(bound, np.uint8)
    offs = np.zeros((n + 255) // 256 + 1, np.uint64)
    back = np.zeros(n, a.dtype)
    total = C.c_uint64(0)
    def run(a, out, offs, back):
        best = [None, None]
        for _ in range(3):
            t0 = time.perf_counter()
            rc = lib.oroch_group_encode_host(code, a.ctypes.data, n, out.ctypes.data, bound,
                                             offs.ctypes.data, C.byref(total))
            t1 = time.perf_counter()
            rc = rc or lib.oroch_group_decode_host(code, out.ctypes.data, offs.ctypes.data, n,
                                                   back.ctypes.data)
            t2 = time.perf_counter()
            if rc:
                raise SystemExit("host entry points failed: %d" % rc)
            best[0] = min(best[0] or 1e9, t1 - t0)
            best[1] = min(best[1] or 1e9, t2 - t1)
        if not np.array_equal(back, a):
            raise SystemExit("host round trip mismatch")
        return best

    best = run(a, out, offs, back)
    # the same with the caller's buffers pinned: no staging copy on the host side
    import torch

    pa = torch.from_numpy(a).pin_memory()
    pout = torch.zeros(bound, dtype=torch.uint8).pin_memory()
    pback = torch.zeros(n, dtype=pa.dtype).pin_memory()
    pinned = run(pa.numpy(), pout.numpy(), offs, pback.numpy())
    # what the link carries one way at a time: plain copies of the same pinned buffers
    link = {}
    dev_copy = torch.empty_like(x_dev[:n])
    for name, dst, src in (("h2d", dev_copy, pa), ("d2h", pback, dev_copy)):
        t = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            t = min(t, time.perf_counter() - t0)
        link[name] = a.nbytes / t / 1e9
    return {"values": n, "encode_GBps": a.nbytes / best[0] / 1e9, "decode_GBps": a.nbytes / best[1] / 1e9,
            "pinned_encode_GBps": a.nbytes / pinned[0] / 1e9, "pinned_decode_GBps": a.nbytes / pinned[1] / 1e9,
            "link_GBps": link, "compressed_over_raw": int(total.value) / a.nbytes,
            "note": "host buffers in and out through the library's pipe (64 MiB chunks; copy in, kernels and "
                    "copy out of neighbouring chunks overlap; buffers allocated once), uncompressed bytes per "
                    "second, best of 3; pageable buffers are staged through pinned memory, pinned ones go "
                    "as they are; encode moves 1 + compressed_over_raw bytes over the link per byte, the "
                    "larger share host to device, decode the other way round"}


def kernel_ms(fn, steps):
    """Device time of the library's kernels per call of fn, summed over the
    kernels it launches (HIP events on the launch stream)."""
    import torch

    import oroch_amd

    fn()
    torch.cuda.synchronize()
    oroch_amd.profile_start(64 * (steps + 1))
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    prof = oroch_amd.profile_stop()
    return {k: ms / cnt for k, (ms, cnt) in prof.items()}


def other_configs(device, steps=5, shrink=0):
    """The other BASELINE configurations at their full sizes on this GPU: kernel
    time per direction, algorithmic bytes (values once + code once) and the
    fraction of 8 TB/s that makes; every one round-trip-checked."""
    import torch

    import oroch_amd as hip

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import devgen

    out = {}
    size = lambda log2n: 1 << max(12, log2n - shrink)  # (smaller for the tests of this script)

    def entry(name, raw, coded, enc, dec, extra=None):
        e = {"values_bytes": raw, "code_bytes": coded, "algorithmic_bytes": raw + coded,
             "encode_kernel_ms": enc, "decode_kernel_ms": dec,
             "encode_ms": sum(enc.values()), "decode_ms": sum(dec.values())}
        for d in ("encode", "decode"):
            e[d + "_frac"] = (raw + coded) / (e[d + "_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        e["round_trip"] = "ok"
        if extra:
            e.update(extra)
        out[name] = e

    def group(name, x, np_dtype):
        n = x.numel()
        enc = hip.GroupEncoder(np_dtype, n, device)
        back = torch.empty_like(x)
        code = hip.dtype_code(np_dtype)

        def decode():
            rc = hip.lib.oroch_group_decode(code, enc.out.data_ptr(), enc.offsets.data_ptr(), n, back.data_ptr(),
                                            hip.capi._stream())
            if rc:
                raise SystemExit("decode failed: %d" % rc)
        e = kernel_ms(lambda: enc.encode(x), steps)
        d = kernel_ms(decode, steps)
        if not torch.equal(back.view(torch.uint8), x.view(torch.uint8)):
            raise SystemExit("round trip mismatch in " + name)
        total = int(enc.offsets[-1].item())
        entry(name, x.numel() * x.element_size(), total, e, d, {"compressed_over_raw": total / (n * x.element_size())})

    # configuration 2: int32, 2^26 values, zigzag bit-packing and frame of reference
    n = size(26)
    x = devgen.uniform_bits(n, torch.int32, 2, 16, device, shift=-20000 + 1000000)
    lo, hi = int(x.min()), int(x.max())
    zz = max((lo << 1) ^ (lo >> 31), (hi << 1) ^ (hi >> 31)) & 0xffffffff
    nb_pck, nb_for = zz.bit_length(), (hi - lo).bit_length()
    keep = {}
    e = kernel_ms(lambda: keep.__setitem__("p", hip.bitpck_encode(x, nb_pck)), steps)
    d = kernel_ms(lambda: keep.__setitem__("b", hip.bitpck_decode(keep["p"], n, nb_pck, np.int32)), steps)
    if not torch.equal(keep["b"], x):
        raise SystemExit("round trip mismatch in bitpck")
    entry("config2 bitpck int32 2^26 (zigzag, %d bits)" % nb_pck, 4 * n, keep["p"].numel(), e, d)
    e = kernel_ms(lambda: keep.__setitem__("p", hip.bitfor_encode(x, lo, nb_for)), steps)
    d = kernel_ms(lambda: keep.__setitem__("b", hip.bitfor_decode(keep["p"], n, lo, nb_for, np.int32)), steps)
    if not torch.equal(keep["b"], x):
        raise SystemExit("round trip mismatch in bitfor")
    entry("config2 bitfor int32 2^26 (origin = min, %d bits)" % nb_for, 4 * n, keep["p"].numel(), e, d)
    keep.clear()
    del x

    # configuration 3: uint32, 2^28 values, 1 % outliers, one patched code over the whole array
    n = size(28)
    xs = devgen.outlier_mix(n, torch.int32, 3, device)
    xu = xs.view(torch.uint32)
    e = kernel_ms(lambda: keep.__setitem__("c", hip.bitpfr_encode(xu, 0, 10, capacity=n // 64)), steps)
    b, idx, val, nexc = keep["c"]
    d = kernel_ms(lambda: keep.__setitem__("b", hip.bitpfr_decode(b, n, 0, 10, idx, val, np.uint32)), steps)
    if not torch.equal(keep["b"].view(torch.int32), xs):
        raise SystemExit("round trip mismatch in bitpfr")
    entry("config3 bitpfr uint32 2^28 (origin 0, 10 bits, whole array)", 4 * n, b.numel() + 16 * int(nexc), e, d,
          {"exceptions": int(nexc)})
    keep.clear()
    del b, idx, val
    group("config3 per-block selection uint32 2^28", xu, np.uint32)
    del xs, xu

    # configuration 5: int16 and uint64 through integer_group
    x = devgen.outlier_mix(size(28), torch.int16, 5, device, body_bits=9, out_bits=16, permille=5, shift=-256)
    group("config5 int16 2^28 (9-bit noise + 0.5% full range)", x, np.int16)
    del x
    x = devgen.outlier_mix(size(27), torch.int64, 6, device, body_bits=20, out_bits=50, shift=1 << 40).view(torch.uint64)
    group("config5 uint64 2^27 (20 bits above 2^40 + 1% 50-bit outliers)", x, np.uint64)
    del x

    # configuration 4: int64, 2^30 values, eight block shapes in turn
    x = devgen.block_mix_int64(size(30), 4, device)
    group("config4 int64 2^30 (eight block shapes in turn)", x, np.int64)
    del x
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=30,
                    help="log2 of values per GPU (weak scaling) or of the whole array (strong scaling)")
    ap.add_argument("--config", choices=("headline", "int64mix"), default="headline",
                    help="headline: int32, 10-bit body + 1%% 28-bit outliers; int64mix: int64, eight block "
                         "shapes in turn (constant, packed, patched, varint, far origin, raw, ramp, 3-bit)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 2^log2n values per GPU; strong: one array of 2^log2n values cut into "
                         "contiguous block ranges, one per GPU")
    ap.add_argument("--cpu-log2n", type=int, default=28)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="leave out the other BASELINE configurations (the default one-GPU run times them too)")
    ap.add_argument("--other-shrink", type=int, default=0,
                    help="the other configurations at 2^-K of their sizes (tests of this script)")
    ap.add_argument("--cpu-native-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl is RCCL)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0")
    args = ap.parse_args()

    if args.cpu_native_child:
        # the CPU baseline with whatever OROCH_REF_LIBRARY names; no GPU is touched
        print(json.dumps(cpu_baseline(args.cpu_log2n, 3, args.config, native_child=True)))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # convenience: start one rank per GPU as children and pass their exit code on
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # (child processes first, before this one initialises the GPU)
    native = None
    if world == 1 and not args.no_cpu_baseline:
        native = cpu_native_baseline(args.cpu_log2n, args.config)

    import torch
    import torch.distributed as dist

    import oroch_amd
    from oroch_amd import sharding

    if args.one_device:
        local_rank = 0
    elif world > torch.cuda.device_count():
        # one rank per GPU: ranks that shared a card would time each other
        raise SystemExit("%d ranks but %d GPUs: start one rank per GPU (or pass --one-device to rehearse "
                         "on one card)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rc = oroch_amd.lib.oroch_hip_init(local_rank)
    if rc != 0:
        raise SystemExit("oroch_hip_init failed: %s" % oroch_amd.lib.oroch_hip_strerror(rc).decode())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    seed = 3
    if args.scaling == "strong":
        # one array, rank r owns the block range shard_range gives it
        start, stop = sharding.shard_range(1 << args.log2n, rank, world)
        n = stop - start
    else:
        # rank r owns values [r*n, (r+1)*n) of the N*n-value array
        n = 1 << args.log2n
        start = rank * n
    if args.config == "int64mix":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import devgen

        np_dtype, width = np.int64, 8
        x = devgen.block_mix_int64(n, seed + 1, device, start=start)
    else:
        np_dtype, width = np.int32, 4
        x = gen_outlier_mix_torch(n, seed, device, start=start)
    enc = oroch_amd.GroupEncoder(np_dtype, n, device)
    back = torch.empty_like(x)
    code = oroch_amd.dtype_code(np_dtype)

    def encode():
        enc.encode(x)

    def decode():
        rc = oroch_amd.lib.oroch_group_decode(code, enc.out.data_ptr(), enc.offsets.data_ptr(), n,
                                              back.data_ptr(), oroch_amd.capi._stream())
        if rc:
            raise SystemExit("decode failed: %d" % rc)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        encode()
        decode()
    barrier()
    total = int(enc.offsets[-1].item())
    if not torch.equal(back, x):
        raise SystemExit("round trip mismatch on rank %d" % rank)

    oroch_amd.profile_start(16 * (args.steps + 1))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    enc_ms = dec_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev[0].record()
        encode()
        ev[1].record()
        decode()
        ev[2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = oroch_amd.profile_stop()
    mine = {"rank": rank, "device": local_rank,
            "uuid": str(getattr(torch.cuda.get_device_properties(local_rank), "uuid", "")),
            "name": torch.cuda.get_device_name(local_rank), "values": n, "ms_per_step": elapsed / args.steps * 1e3}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    elapsed = sharding.max_over_ranks(elapsed, device)
    # per-phase device time of the last step (events are reused, so only the last survives)
    enc_ms = ev[0].elapsed_time(ev[1])
    dec_ms = ev[1].elapsed_time(ev[2])

    counts = sharding.gather_byte_counts(total, device)
    # host to device to host: every rank drives its own GPU's pipe at the same time
    # (after the timed region; the ranks share the host's memory bandwidth and PCIe roots)
    e2e = None
    if not args.no_end_to_end:
        barrier()
        e2e = end_to_end(x, min(n, 1 << 27))
        if world > 1:
            keys = ("encode_GBps", "decode_GBps", "pinned_encode_GBps", "pinned_decode_GBps")
            mine = torch.tensor([e2e[k] for k in keys], dtype=torch.float64, device=device)
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
            for k, v in zip(keys, mine.tolist()):
                e2e[k] = v
            e2e["note"] = "sum over %d ranks running at the same time, one GPU and one pipe each; " % world \
                          + e2e["note"]
    if rank == 0:
        raw = n * width  # rank 0's share; every rank holds as much (the last of a strong split maybe less)
        ms_per_step = elapsed / args.steps * 1e3
        values_all = (1 << args.log2n) if args.scaling == "strong" else world * n
        value = 2 * values_all * width / (elapsed / args.steps) / 1e9
        kern = {}
        for name, (ms, cnt) in prof.items():
            kern[name] = ms / cnt
        # algorithmic HBM bytes per launch: encode side reads the values once and
        # writes the stream; decode reads the stream and writes the values
        algo = {"select": raw, "encode": raw + total, "decode": raw + total, "scan": 0}
        enc_path_ms = sum(kern.get(k, 0.0) for k in ("select", "scan", "encode"))
        dec_path_ms = kern.get("decode", 0.0)
        paths = {
            "encode": {"ms": enc_path_ms, "algorithmic_bytes": raw + total,
                       "kernels": [k for k in ("select", "scan", "encode") if k in kern]},
            "decode": {"ms": dec_path_ms, "algorithmic_bytes": raw + total, "kernels": ["decode"]},
        }
        for d, p in paths.items():
            p["achieved"] = p["algorithmic_bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] else 0.0
            p["frac"] = p["achieved"] / HBM_PEAK_GBS
            p["traffic"], p["traffic_record"] = (measured_traffic(d, args.log2n) if args.config == "headline"
                                                 and n == 1 << 30 else (None, None))
        # the figure of merit is the slower direction
        worse = min(paths, key=lambda d: paths[d]["frac"])
        line = {
            "metric": "device-resident encode+decode GB/s (uncompressed bytes), per-block "
                      + ("bitpfr int32" if args.config == "headline" else "selection int64 mix"),
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "int32" if width == 4 else "int64",
            "data": "synthetic",
            "config": {"workload": ("int32 2^%d values %s, 10-bit body + 1%% 28-bit outliers, shift -200, "
                                    if args.config == "headline" else
                                    "int64 2^%d values %s, eight block shapes in turn (constant, 12-bit, "
                                    "12-bit + 1%% 40-bit outliers, signed geometric, far origin, raw, ramp, "
                                    "3-bit), ") % (args.log2n, "per GPU" if args.scaling == "weak"
                                                  else "in all, cut into one block range per GPU")
                                   + "blocks of 256, select+scan+encode then decode",
                       "values_per_gpu": n, "block_values": 256,
                       "compressed_bytes_per_gpu": total,
                       "compressed_over_raw": total / raw, "parallelism": "blocks sharded over %d GPU(s), "
                       "no data-path collective" % world},
            "encode_GBps": raw / (enc_ms * 1e-3) / 1e9, "decode_GBps": raw / (dec_ms * 1e-3) / 1e9,
            "kernel_ms": kern,
            "roofline": {"bound": "hbm", "kernel": "%s direction (%s)" % (worse, "+".join(paths[worse]["kernels"])),
                         "achieved": paths[worse]["achieved"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": paths[worse]["frac"],
                         "traffic": paths[worse]["traffic"],
                         "traffic_record": paths[worse]["traffic_record"],
                         "algorithmic_bytes_per_launch": paths[worse]["algorithmic_bytes"],
                         "encode_path": paths["encode"], "decode_path": paths["decode"]},
            "shard_bytes": counts,
            "ranks": ranks,
            "built_with": oroch_amd.lib.oroch_hip_build_info().decode(),
        }
        if world == 1 and args.config == "headline" and not args.no_other_configs:
            del x, back, enc
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(device, shrink=args.other_shrink)
        if e2e is not None:
            line["end_to_end"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_log2n, seed, args.config, native=native)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

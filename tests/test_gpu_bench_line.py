"""bench.py prints one JSON line with the fields the benchmark contract names."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_complete_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log2n", "22", "--steps", "2",
                          "--warmup", "1", "--cpu-log2n", "18", "--other-shrink", "8"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "GB/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "int32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    # the host is named, and the reference's whole-sequence codecs are timed beside the blocks
    if c["kind"] == "reference":
        assert c["cpu_model"] and c["nproc"] >= c["cores"]
        assert set(k.split()[0] for k in c["whole_array_1thread_GBps"]) == {"bitpck", "bitfor", "bitpfr"}
        assert "O3_march_native" in c
    # which compiler built the kernels, and every rank's device
    assert "clang" in d["built_with"]
    assert len(d["ranks"]) == 1 and d["ranks"][0]["rank"] == 0 and d["ranks"][0]["ms_per_step"] > 0
    # a traffic figure is only quoted for the kernels it was measured on
    assert r["traffic"] is None
    # the other configurations: every one round-tripped, with both directions priced
    o = d["other_configs"]
    assert len(o) == 7 and all(v["round_trip"] == "ok" for v in o.values())
    for name, v in o.items():
        assert 0 < v["encode_frac"] < 1 and 0 < v["decode_frac"] < 1, name
    for head in ("config2 bitpck", "config2 bitfor", "config3 bitpfr", "config3 per-block", "config4 int64",
                 "config5 int16", "config5 uint64"):
        assert any(k.startswith(head) for k in o), head
    # the figure of merit is the slower of the two directions
    assert abs(r["frac"] - min(r["encode_path"]["frac"], r["decode_path"]["frac"])) < 1e-9
    assert r["kernel"].split()[0] in ("encode", "decode")


@pytest.mark.gpu
def test_bench_strong_scaling_of_the_int64_mix():
    """One int64 array cut into block ranges: alone, and as two ranks that share
    the one GPU of this box (gloo between them; the ranks exchange byte counts only)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "int64mix", "--scaling", "strong",
            "--log2n", "21", "--steps", "2", "--warmup", "1", "--cpu-log2n", "16"]
    lines = {}
    for world, extra in ((1, []), (2, ["--gpus", "2", "--one-device", "--backend", "gloo"])):
        env = dict(os.environ, MASTER_PORT="29533")
        out = subprocess.run(base + extra, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        got = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(got) == 1, out.stdout[-2000:]
        lines[world] = json.loads(got[0])
    for world, d in lines.items():
        assert d["n_gpus"] == world and d["scaling"] == "strong" and d["dtype"] == "int64"
        assert "int64" in d["metric"] and "int64 2^21 values in all" in d["config"]["workload"]
        assert len(d["shard_bytes"]) == world and d["value"] > 0
    # the shards of the two ranks together are the stream of the one
    assert sum(lines[2]["shard_bytes"]) == lines[1]["shard_bytes"][0]
    assert lines[2]["config"]["values_per_gpu"] == (1 << 20)


@pytest.mark.gpu
def test_bench_strong_scaling_of_the_headline_and_one_rank_per_gpu():
    """The headline array cut into block ranges for two ranks (sharing this box's GPU
    on request), every rank's device and time in the line; without the request,
    more ranks than GPUs is refused before anything is timed."""
    import torch

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong", "--log2n", "21", "--steps", "2",
            "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end", "--gpus", "2", "--backend", "gloo"]
    env = dict(os.environ, MASTER_PORT="29534")
    out = subprocess.run(base + ["--one-device"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dtype"] == "int32"
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["values"] == 1 << 20 for r in d["ranks"])
    assert all(r["device"] == 0 and r["ms_per_step"] > 0 for r in d["ranks"])
    assert d["ms_per_step"] >= max(r["ms_per_step"] for r in d["ranks"]) - 1e-6
    if torch.cuda.device_count() < 2:
        env = dict(os.environ, MASTER_PORT="29535")
        out = subprocess.run(base, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode != 0
        assert "one rank per GPU" in out.stderr + out.stdout

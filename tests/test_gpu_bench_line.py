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
                          "--warmup", "1", "--cpu-log2n", "18"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
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

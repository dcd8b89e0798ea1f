"""Where the reference tree is present, its bench programs (tests/bench/*.cc) must
compile as they are against include/oroch/*.h: the headers are a drop-in for the
reference's.  (Its unit tests are compiled the same way by oracle/Makefile and run
on the GPU by test_gpu_reference_units.py.)"""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests", "bench")), reason="no reference tree here")
def test_reference_bench_programs_compile_against_these_headers():
    sources = sorted(glob.glob(os.path.join(REF, "tests", "bench", "*.cc")))
    assert len(sources) >= 3
    for src in sources:
        out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (src, out.stderr[-2000:])

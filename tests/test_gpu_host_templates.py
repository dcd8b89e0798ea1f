"""The C++ host templates of include/oroch/ (reference class names over the
C ABI), exercised by a prebuilt test binary that checks every result against
the CPU oracle.  Built by `make -C tests/cxx` (done by __graft_entry__.build)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "cxx", "host_templates_test")


def test_cxx_host_templates():
    assert os.path.exists(BINARY), "build it first: make -C tests/cxx"
    r = subprocess.run([BINARY], capture_output=True, text=True, timeout=600)
    tail = (r.stdout[-2000:] + "\n" + r.stderr[-4000:])
    assert r.returncode == 0, tail
    assert " 0 failed" in r.stdout.splitlines()[-1], tail

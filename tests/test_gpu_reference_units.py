"""The reference's own unit tests (tests/unit/*.cc: bitblk, bitpck, bitfor, bitpfr,
integer_array, integer_codec, integer_group, normal, offset, varint, zigzag),
compiled unchanged against this repository's include/oroch/*.h and linked with the
HIP library (oracle/Makefile writes oracle/_ref/ref_unit_tests where the reference
tree exists).  Every device call of the headers runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "ref_unit_tests")


@pytest.mark.gpu
def test_reference_unit_tests_pass_on_the_device_path():
    if not os.path.exists(BINARY):
        pytest.skip("oracle/_ref/ref_unit_tests not built (needs the reference tree: make -C oracle)")
    out = subprocess.run([BINARY], cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "All tests passed" in out.stdout, tail

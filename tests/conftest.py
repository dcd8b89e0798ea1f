import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    z = np.load(path, allow_pickle=False)
    doc = json.loads(bytes(z["cases"]).decode())
    return z, doc["cases"], doc["meta_fields"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle as po

    if po.oracle is None:
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"),
                               os.path.join(ROOT, "oracle", "liboroch_oracle.so")])
        po.oracle = po._load(os.path.join(po.HERE, "liboroch_oracle.so"), "orc_")
    return po.oracle

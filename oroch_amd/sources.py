"""Digest of the kernel sources the shipped library is built from.

Measurements kept under profiles/ (HBM traffic from rocprofv3 counter passes,
which cannot run inside the benchmark's own process) record this digest, and
bench.py only quotes them while it still matches: a figure measured on other
kernels is not this build's figure.  (The GPU box has no .git to ask for a
commit, the sources travel with the tree.)
"""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def digest():
    """First 16 hex digits of the SHA-256 over csrc/*.hip, csrc/*.hpp and the C header."""
    h = hashlib.sha256()
    csrc = os.path.join(HERE, "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    paths = [os.path.join(csrc, f) for f in names] + [os.path.join(HERE, "..", "include", "oroch_hip.h")]
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]

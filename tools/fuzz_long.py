"""Differential fuzz of the block path against the oracle for a given time:
python3 tools/fuzz_long.py [seconds] [first seed]
Every round draws an element type, a block count and blocks from the shapes of
tests/datagen.py plus shapes that select varint / varfor and dense exception
lists, and compares descriptors' sizes, offsets, stream bytes and both decoders.
Prints one line per round; stops at the first difference with the seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oroch_amd, datagen as dg
from oracle import pyoracle as po

oracle = po.oracle
assert oracle is not None, "make -C oracle first"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)


def special(rng, dt, n):
    dt = np.dtype(dt)
    W = dt.itemsize * 8
    u = np.dtype("u%d" % dt.itemsize)
    kind = int(rng.integers(0, 4))
    if kind == 0:  # seven-bit values and a few wide ones (varint), maybe away from zero (varfor)
        v = rng.integers(0, 1 << 7, n).astype(np.int64)
        k = int(rng.integers(1, 6))
        v[rng.integers(0, n, k)] = rng.integers(1 << (W - 3), 1 << (W - 1 if W < 64 else 62), k)
        if rng.integers(0, 2):
            v = v + int(rng.integers(0, 1 << (W - 2 if W < 64 else 60)))
    elif kind == 1:  # nearly every value an exception of a narrow body
        b = int(rng.integers(4, max(5, W - 6)))
        v = rng.integers(0, 1 << min(62, b + int(rng.integers(1, 6))), n).astype(np.int64)
    elif kind == 2:  # a width from 1 to 3 bits
        v = rng.integers(0, 1 << int(rng.integers(1, 4)), n).astype(np.int64) + int(rng.integers(0, 50))
    else:  # 17 to 40 exceptions
        b = int(rng.integers(4, max(5, W - 6)))
        v = rng.integers(0, 1 << b, n).astype(np.int64)
        k = int(rng.integers(17, 41))
        v[rng.integers(0, n, k)] = rng.integers(0, 1 << min(62, W - 1), k)
    with np.errstate(over="ignore"):
        return np.ascontiguousarray(v.astype(np.uint64).astype(u).view(dt))


t0 = time.time()
rounds = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    dt = dg.DTYPES[int(rng.integers(0, len(dg.DTYPES)))]
    nblocks = int(rng.integers(1, 1500))
    mix = float(rng.random())
    blocks = [special(rng, dt, 256) if rng.random() < mix else dg.fuzz_block(rng, dt, 256) for _ in range(nblocks)]
    if rng.integers(0, 2):
        blocks.append(dg.fuzz_block(rng, dt, int(rng.integers(1, 256))))
    a = np.concatenate(blocks)
    exp, exp_offs = oracle.blocks_encode(a, 256, 16)
    x = torch.from_numpy(a).to(dev)
    # (now and then through the one-block kernels, which the 64-bit types always use)
    one = [int(rng.integers(0, 3) == 0), int(rng.integers(0, 3) == 0)]
    with oroch_amd.option("select_one_block", one[0]), oroch_amd.option("encode_one_block", one[1]):
        stream, offs = oroch_amd.group_encode(x)
    ok = np.array_equal(offs.cpu().numpy().astype(np.uint64), exp_offs) and np.array_equal(stream.cpu().numpy(), exp)
    back = oroch_amd.group_decode(stream, offs, a.size, a.dtype)
    ok = ok and np.array_equal(back.cpu().numpy().view(a.dtype), a)
    with oroch_amd.option("decode_one_block", 1):
        back1 = oroch_amd.group_decode(stream, offs, a.size, a.dtype)
    ok = ok and torch.equal(back1, back)
    rounds += 1
    if not ok:
        print("MISMATCH seed", seed, a.dtype, nblocks, flush=True)
        sys.exit(1)
    if rounds % 20 == 0:
        print("round %d seed %d %s %d blocks ok (%.0f s)" % (rounds, seed, a.dtype, nblocks, time.time() - t0), flush=True)
    seed += 1
print("%d rounds, seeds up to %d: all equal" % (rounds, seed - 1))

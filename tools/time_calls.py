"""Per-call cost of the single-value array operations (find, at, insert at the end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oroch_amd
a = oroch_amd.IntegerArray(np.int32)
a.append(np.arange(100000, dtype=np.int32))
for name, fn in (("find", lambda i: a.find(np.int32(i * 7))), ("at", lambda i: a.at(i * 7)),
                 ("insert at the end", lambda i: a.insert(len(a), i))):
    for i in range(50): fn(i)
    t0 = time.perf_counter()
    for i in range(2000): fn(i)
    print("%-18s %.1f us per call" % (name, (time.perf_counter() - t0) / 2000 * 1e6))

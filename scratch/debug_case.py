import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, datagen as dg
from oracle import pyoracle as po
z = np.load("tests/golden/golden_v1.npz")
cases = json.loads(bytes(z["cases"]).decode())["cases"]
bad = 0
for i, c in enumerate(cases):
    if c["kind"] != "group" or c["n"] == 0: continue
    a = z["c%d_in" % i].view(dg.DTYPES[c["dtype"]])
    x = torch.from_numpy(a).cuda()
    s, o = oroch_amd.group_encode(x)
    back = oroch_amd.group_decode(s, o, a.size, a.dtype).cpu().numpy()
    if not np.array_equal(back, a):
        bad += 1
        if bad <= 6:
            m = po.oracle.select(a)
            w = np.nonzero(back != a)[0]
            print(i, c["dtype"], c["n"], m, "wrong at", w[:8], "got", back[w[:8]], "want", a[w[:8]])
            print("   stream tail", s.cpu().numpy()[-24:])
print("bad", bad)

import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, datagen as dg
z = np.load("tests/golden/golden_v1.npz")
cases = json.loads(bytes(z["cases"]).decode())["cases"]
dbg = torch.zeros(64, dtype=torch.int32, device="cuda")
if hasattr(oroch_amd.lib, "oroch_debug_set"):
    oroch_amd.lib.oroch_debug_set(ctypes.c_void_p(dbg.data_ptr()))
names = "tag nv nb doff nout inb vnb ibase ilen vbase len gp0 val0 epos0 hi0 FULL".split()
for i in (132, 186):
    c = cases[i]
    a = z["c%d_in" % i].view(dg.DTYPES[c["dtype"]])
    x = torch.from_numpy(a).cuda()
    s, o = oroch_amd.group_encode(x)
    dbg.zero_()
    back = oroch_amd.group_decode(s, o, a.size, a.dtype).cpu().numpy()
    torch.cuda.synchronize()
    print("case", i, "ok" if np.array_equal(back, a) else "BAD", dict(zip(names, dbg.cpu().numpy()[:16].tolist())), flush=True)

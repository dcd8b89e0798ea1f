"""Times searching and random access over the block images of the headline array
(int32, 2^28 values, 10-bit noise with 1 % outliers around -200): one value that is
nowhere (every block read), one value the widths rule out, a batch of values, and
batches of random positions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oroch_amd, devgen
from oroch_amd import capi
dev = torch.device("cuda", 0)

def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

n = 1 << 28
x = devgen.outlier_mix(n, torch.int32, 7, dev, shift=-200)
stream, offs = oroch_amd.group_encode(x)
sb = stream.numel()
code = capi.dtype_code(np.int32)
st = capi._stream()

def finder(values):
    nd = capi._raw_needles(values, code, dev)
    out = torch.empty(nd.numel(), dtype=torch.int64, device=dev)
    def go():
        capi._check(capi.lib.oroch_group_find(code, stream.data_ptr(), offs.data_ptr(), n, nd.data_ptr(), nd.numel(), out.data_ptr(), st))
    return go, out

go, out = finder([300])        # within the noise band but (checked below) maybe present
ms = timed(go); print("find one value inside the band: %.3f ms, first at %d" % (ms, out.cpu().numpy().view(np.uint64)[0]))
absent = 5_000_000              # wider than any body, narrower than the outliers: every block decoded
go, out = finder([absent])
ms = timed(go); print("find one absent value (exception-wide): %.3f ms -> %.0f GB/s of stream, %.1f%% of 8 TB/s; result %x" % (ms, sb / ms / 1e6, sb / ms / 1e6 / 80, out.cpu().numpy().view(np.uint64)[0]))
go, out = finder([-2_000_000_000])
ms = timed(go); print("find one absent value below every origin: %.3f ms -> %.0f GB/s of stream" % (ms, sb / ms / 1e6))
go, out = finder(np.arange(5_000_000, 5_000_016))
ms = timed(go); print("find 16 absent values: %.3f ms" % ms)
for m in (1 << 10, 1 << 16, 1 << 20, 1 << 24):
    idx = torch.randint(0, n, (m,), device=dev)
    o = torch.empty(m, dtype=torch.int32, device=dev)
    def go():
        capi._check(capi.lib.oroch_group_at(code, stream.data_ptr(), offs.data_ptr(), n, idx.data_ptr(), m, o.data_ptr(), st))
    ms = timed(go)
    assert torch.equal(o, x[idx])
    print("at: %8d random positions: %.3f ms -> %.1f M positions/s" % (m, ms, m / ms / 1e3))
dec = timed(lambda: oroch_amd.group_decode(stream, offs, n, np.int32))
print("decode everything: %.3f ms" % dec)

"""Per-phase timeline of a conv workgroup from in-kernel s_memtime stamps (diagnostic build lib_stamps.so, selected with
HN_LIB_PATH; built by the recipe in the commit message / DESIGN).  Phases: entry -> index math -> first tile landed ->
fragments loaded -> k loop -> epilogue issued -> stores retired.  usage: stamps.py tile n h w cin cout r"""
import ctypes as C, sys, os
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import numpy as np, torch
from hn_amd import ops, _lib
from hn_amd.weights import split_f16x3
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)
tile, n, h, w, cin, cout, r = map(int, sys.argv[1:8])
res = len(sys.argv) > 8
g = torch.Generator().manual_seed(0)
x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
wt = torch.randn((cout, r, r, cin), generator=g) * 0.03
w16 = split_f16x3(wt).cuda(); wt = wt.cuda(); b = torch.randn((cout,), generator=g).cuda()
resid = ops.to_split(torch.randn((n, h, w, cout), generator=g).cuda()) if res else None
for _ in range(5):
    y = ops.conv2d_nhwc(x, wt, b, pad=r // 2, relu=True, w16=w16, out_split=True, tile=tile, residual=resid)
torch.cuda.synchronize()
lib = _lib.load()
nb = min(8192, 8192)
buf = (C.c_ulonglong * (8 * nb))()
lib.hn_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.hn_debug_read_stamps(buf, 8 * nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
a = a[a[:, 0] > 0]
d = np.diff(a[:, :7], axis=1)
import os
names = ["index math", "first tile (DMA latency)", "frag preload", "k loop", "epilogue issue", "store drain"] if not os.environ.get("EPI") else ["barrier", "pass0 write+wait", "pass0 compute+store", "pass1 write+wait", "pass1 compute+store", "store drain"]
print(f"tile {tile} {n}x{h}x{w}x{cin}->{cout} r{r} residual={res}: {len(a)} workgroups stamped; median cycles per phase")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {np.median(d[:, i]):9.0f}   (p10 {np.percentile(d[:, i], 10):.0f}, p90 {np.percentile(d[:, i], 90):.0f})")
print(f"  {'total':26s} {np.median(a[:, 6] - a[:, 0]):9.0f}")
print(f"  {'span over all workgroups':26s} {int(a[:, 6].max() - a[:, 0].min()):9d}   (first entry -> last store retired, only meaningful when the stamp buffer holds ONE launch; entry spread {int(a[:, 0].max() - a[:, 0].min())})")

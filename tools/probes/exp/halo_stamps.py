"""Phase stamps of conv3x3_halo_kernel (HN_HALO_STAMPS=1): load wait / k loop / epilogue, median cycles per workgroup."""
import ctypes as C, os, sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
os.environ["HN_HALO_STAMPS"] = "1"
import numpy as np, torch
from hn_amd import ops, _lib
from hn_amd.weights import split_f16x3
n, h, w, cin = 32, 200, 272, 64
res = len(sys.argv) > 1
g = torch.Generator().manual_seed(0)
x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
wt = torch.randn((64, 3, 3, cin), generator=g) * 0.03
w16 = split_f16x3(wt).cuda(); wt = wt.cuda(); b = torch.randn((64,), generator=g).cuda()
resid = ops.to_split(torch.randn((n, h, w, 64), generator=g).cuda()) if res else None
t = ops.HipTimer()
for _ in range(3):
    ops.conv2d_nhwc(x, wt, b, pad=1, relu=True, w16=w16, out_split=True, residual=resid)
torch.cuda.synchronize()
t.start()
for _ in range(10):
    y = ops.conv2d_nhwc(x, wt, b, pad=1, relu=True, w16=w16, out_split=True, residual=resid)
t.stop()
print(f"{t.elapsed_ms() * 100:.1f} us per launch")
lib = _lib.load()
buf = (C.c_ulonglong * (4 * 8192))()
lib.hnx_debug_halo_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.hnx_debug_halo_stamps(buf, 4 * 8192) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
a = a[a[:, 0] > 0]
d = np.diff(a, axis=1)
for i, nm in enumerate(["prologue + first loads", "k loop (18 taps)", "epilogue + drain"]):
    print(f"  {nm:26s} {np.median(d[:, i]):9.0f}   (p10 {np.percentile(d[:, i], 10):.0f}, p90 {np.percentile(d[:, i], 90):.0f})")
print(f"  total {np.median(a[:, 3] - a[:, 0]):.0f}   residual={res}")

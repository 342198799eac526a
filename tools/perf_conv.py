"""Time one conv shape (development aid).  usage: perf_conv.py prec tile n h w cin cout r [stride dil iters affine]"""
import sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops
from hn_amd.weights import split_f16x3
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

a = sys.argv[1:]
prec, tile = a[0], int(a[1])
n, h, w, cin, cout, r = map(int, a[2:8])
stride = int(a[8]) if len(a) > 8 else 1
dil = int(a[9]) if len(a) > 9 else 1
iters = int(a[10]) if len(a) > 10 else 20
affine = int(a[11]) if len(a) > 11 else 0
pad = dil * (r // 2)
g = torch.Generator().manual_seed(0)
x = torch.randn((n, h, w, cin), generator=g).cuda()
wt = (torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5).cuda()
b = torch.randn((cout,), generator=g).cuda()
w16 = split_f16x3(wt.cpu()).cuda() if prec == "f16x3" else None
sc = sh = None
if affine:
    sc = torch.rand((n, cin), generator=g).cuda() + 0.5
    sh = torch.randn((n, cin), generator=g).cuda() * 0.1
oh, ow = ops.conv_out_size(h, w, r, r, stride, pad, dil)
y = torch.empty((n, oh, ow, cout), device="cuda")
osplit = int(a[12]) if len(a) > 12 else 0
if prec == "f16x3":
    x = ops.to_split(x, sc, sh, relu=bool(affine))   # S32 input, produced once (as the engines do)
    sc = sh = None
    if osplit:
        y = torch.empty((n, oh, ow, cout // 32, 2, 32), device="cuda", dtype=torch.float16)
import os
kw = dict(stride=stride, pad=pad, dil=dil, relu=True, tile=tile, w16=w16, out=y, in_scale=sc, in_shift=sh)
if prec == "f16x3":
    kw["splitk"] = os.environ.get("HN_SPLITK", "1") != "0"
for _ in range(3):
    ops.conv2d_nhwc(x, wt, b, **kw)
torch.cuda.synchronize()
# shader clock (in-kernel sampler on a side stream) and socket power (rocm-smi) WHILE the loop runs
import subprocess, threading
side = torch.cuda.Stream()
clock = torch.zeros((1,), device="cuda")
power = []
def _smi():
    import time
    time.sleep(0.25)
    try:
        out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        power.extend(l.split(":")[-1].strip() for l in out.splitlines() if "Socket Graphics" in l)
    except Exception:
        pass
th = threading.Thread(target=_smi)
if iters >= 400:
    th.start()
t = ops.HipTimer()
t.start()
for i in range(iters):
    if i == iters // 4:
        ops.clock_sample(200000 if iters >= 400 else 2000, out=clock, stream=side)
    ops.conv2d_nhwc(x, wt, b, **kw)
t.stop()
ms = t.elapsed_ms() / iters
if iters >= 400:
    th.join()
fl = 2.0 * n * oh * ow * cout * r * r * cin
pw = power[0] if power else "?"
print(f"{prec} tile={tile} {n}x{h}x{w}x{cin}->{cout} r{r} s{stride} d{dil} aff{affine} osplit{osplit}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  clock {float(clock.item()):.0f} MHz  power {pw} W")

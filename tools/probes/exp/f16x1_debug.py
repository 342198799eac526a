"""f16x1 (terms = 1) vs the fp16-operand reference (development aid)."""
import os, sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd")); sys.path.insert(0, str(R))
import torch
from hn_amd import ops
from hn_amd.weights import split_f16x3
from oracle import ops_ref
g = torch.Generator().manual_seed(1)
n, h, w, cin, cout, r = 2, 44, 44, 64, 64, 3
x = torch.randn((n, h, w, cin), generator=g); wt = torch.randn((cout, r, r, cin), generator=g) * 0.05
half = ops_ref.conv2d_nhwc(x.half().double(), wt.half().double(), None, 1, r // 2, 1).float()
xs = ops.to_split(x.cuda()); w16 = split_f16x3(wt).cuda(); wd = wt.cuda()
for rep in range(4):
    with ops.f16_terms(1):
        y = ops.conv2d_nhwc(xs, wd, None, pad=1, w16=w16, tile=1, splitk=False).cpu()
    err = (y - half).abs()
    bad = (err > 1e-3)
    idx = bad.nonzero()
    print(f"rep {rep}: max err {float(err.max()):.3e}, bad elements {int(bad.sum())} of {bad.numel()}; first bad {idx[:3].tolist()}; rows with bad: {sorted(set((idx[:,1]).tolist()))[:12]} cols {sorted(set(idx[:,2].tolist()))[:12]} ch {sorted(set(idx[:,3].tolist()))[:8]}", flush=True)

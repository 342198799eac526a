"""GroupNorm-apply + split pass (hn_affine_split_f32) on the three tower levels at batch 32: time and HBM rate.
HN_SPLIT_GENERIC=1 selects the generic kernel (two 64-bit divisions per item, scale / shift re-read per item)."""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops

g = torch.Generator().manual_seed(0)
for (h, w) in ((100, 136), (50, 68), (25, 34)):
    n, c = 32, 512
    raw = torch.randn((n, h, w, c), generator=g).cuda()
    sc = (torch.rand((n, c), generator=g) + 0.5).cuda(); sh = (torch.randn((n, c), generator=g) * 0.1).cuda()
    act = torch.empty((n, h, w, c // 32, 2, 32), device="cuda", dtype=torch.float16)
    for _ in range(20):
        ops.to_split(raw, sc, sh, relu=True, out=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.to_split(raw, sc, sh, relu=True, out=act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    gb = raw.numel() * 8 / 1e9
    ref = torch.relu(raw * sc[:, None, None, :] + sh[:, None, None, :])
    ok = torch.equal(ops.from_split(act), ops.from_split(ops.to_split(ref)))
    print(f"{n}x{h}x{w}x{c}: {us:.0f} us  {gb/us*1e6/1e3:.2f} TB/s  exact={ok}", flush=True)

"""Can the HBM-bound GroupNorm-apply pass hide under the MFMA-bound (power-capped) tower conv when both run on
different streams?  conv alone, apply alone, both concurrently (same sizes as one tower layer at level 0, batch 32)."""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops
from hn_amd.weights import split_f16x3
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

g = torch.Generator().manual_seed(0)
n, h, w = 32, 100, 136
x = ops.to_split(torch.randn((n, h, w, 256), generator=g).cuda())
wt = (torch.randn((512, 3, 3, 256), generator=g) * 0.03)
w16 = split_f16x3(wt).cuda(); wt = wt.cuda()
raw = torch.randn((n, h, w, 512), generator=g).cuda()
sc = (torch.rand((n, 512), generator=g) + 0.5).cuda(); sh = torch.randn((n, 512), generator=g).cuda() * 0.1
y = torch.empty((n, h, w, 512), device="cuda")
act = torch.empty((n, h, w, 16, 2, 32), device="cuda", dtype=torch.float16)
# argv[1] = priority of the apply stream (0 = default, -1 = high: does the dispatcher then interleave its workgroups with the
# conv's instead of waiting for the conv's queue to drain?)
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
def conv(): ops.conv2d_nhwc(x, wt, None, pad=1, w16=w16, out=y)
def apply(): ops.to_split(raw, sc, sh, relu=True, out=act)
def run(fa, fb, iters=200):
    for _ in range(5):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    for _ in range(iters):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tc, ta, tb = run(conv, None), run(None, apply), run(conv, apply)
print(f"apply stream priority {prio}: ", end="")
print(f"conv alone {tc*1e3:.0f} us   apply alone {ta*1e3:.0f} us   both concurrently {tb*1e3:.0f} us per pair   (serial sum {1e3*(tc+ta):.0f} us)")

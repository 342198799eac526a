"""Does overlapping the HBM-bound passes of one half-batch with the MFMA-bound convolutions of the other half-batch
(two streams, two engine instances) beat one stream at the full batch?  (power-budget filling; round-2 probe)"""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd.fcos_engine import FCOSEngine
from hn_amd.pipeline import HandNetEngine
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

fsd, asd = synth.make_fcos_state_dict(0, 3), synth.make_a2j_state_dict(0)
def mk():
    return HandNetEngine(FCOSEngine(fsd, 3, device="cuda"), A2JEngine(asd, device="cuda"), 3)
rgb, depth = synth.make_rgb(32, seed=1000).cuda(), synth.make_depth(32, seed=2000).cuda()
e0 = mk()
def timeit(fn, steps=15, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / steps
t1 = timeit(lambda: e0.forward_device(rgb, depth))
print(f"one stream, batch 32: {t1*1e3:.2f} ms/step = {32/t1:.0f} frames/s")
for parts in (2, 4):
    engs = [mk() for _ in range(parts)]
    sts = [torch.cuda.Stream() for _ in range(parts)]
    b = 32 // parts
    ins = [(rgb[i*b:(i+1)*b].contiguous(), depth[i*b:(i+1)*b].contiguous()) for i in range(parts)]
    def step():
        for e, s, (x, d) in zip(engs, sts, ins):
            with torch.cuda.stream(s):
                e.forward_device(x, d)
    tp = timeit(step)
    print(f"{parts} streams x batch {b}: {tp*1e3:.2f} ms/step = {32/tp:.0f} frames/s")
    # staggered: stream i starts i/parts of a step later (steady-state interleave of conv and HBM phases)
    del engs

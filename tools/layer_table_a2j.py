"""Per-layer conv timing table of the A2J-only workload (BASELINE config 2: batch 64)."""
import sys
from collections import OrderedDict
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops, synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
a2j = A2JEngine(synth.make_a2j_state_dict(0))
x = synth.make_crops(batch, 176, seed=3000).cuda()
for _ in range(3):
    a2j.forward(x)
torch.cuda.synchronize()
t = ops.HipTimer(); t.start()
for _ in range(20):
    a2j.forward(x)
t.stop()
print(f"# A2J batch {batch}: {t.elapsed_ms()/20:.3f} ms/step = {batch/(t.elapsed_ms()/20)*1e3:.0f} crops/s")
ops.CONV_PROFILE = []
reps = 5
for _ in range(reps):
    a2j.forward(x)
torch.cuda.synchronize()
recs, ops.CONV_PROFILE = ops.CONV_PROFILE, None
tab = OrderedDict()
for kind, macs, timer, shape, *_stage in recs:
    e = tab.setdefault((kind, shape), [0, 0.0, 0.0])
    e[0] += 1; e[1] += timer.elapsed_ms(); e[2] += 2.0 * macs
tot = sum(v[1] for v in tab.values())
print(f"# conv total {tot/reps:.3f} ms/step (event-bracketed, includes launch gaps)")
print(f"{'tile':8s} {'n,h,w,cin,cout,r,stride,dil':40s} {'calls':>5s} {'ms/step':>8s} {'%':>6s} {'TFLOP/s':>8s}")
for (kind, shape), (calls, ms, flop) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(f"{ops.tile_name(kind[1]):11s} {str(shape):40s} {calls/reps:5.0f} {ms/reps:8.3f} {100*ms/tot:6.1f} {flop/ms/1e9:8.1f}")

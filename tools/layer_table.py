"""Per-layer conv timing table (development aid): which shapes / tiles are far from the roof."""
import sys
from collections import OrderedDict
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops, synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd.fcos_engine import FCOSEngine
from hn_amd.pipeline import HandNetEngine
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
fcos = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, precision=prec)
a2j = A2JEngine(synth.make_a2j_state_dict(0), precision=prec)
eng = HandNetEngine(fcos, a2j, 3)
rgb = synth.make_rgb(batch, seed=1000).cuda()
depth = synth.make_depth(batch, seed=2000).cuda()
for _ in range(2):
    eng.forward_device(rgb, depth)
torch.cuda.synchronize()
ops.CONV_PROFILE = []
reps = 3
for _ in range(reps):
    eng.forward_device(rgb, depth)
torch.cuda.synchronize()
recs, ops.CONV_PROFILE = ops.CONV_PROFILE, None
tab = OrderedDict()
for kind, macs, timer, shape, *_stage in recs:
    k = (kind, shape)
    t = tab.setdefault(k, [0, 0.0, 0.0])
    t[0] += 1
    t[1] += timer.elapsed_ms()
    t[2] += 2.0 * macs
tot = sum(v[1] for v in tab.values())
print(f"# precision {prec} batch {batch}: conv total {tot/reps:.2f} ms/step")
print(f"{'kind':10s} {'tile':8s} {'n,h,w,cin,cout,r,stride,dil':42s} {'calls/step':>10s} {'ms/step':>9s} {'%':>6s} {'TFLOP/s':>8s}")
for (kind, shape), (calls, ms, flop) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(f"{kind[0]:10s} {ops.tile_name(kind[1]):11s} {str(shape):42s} {calls/reps:10.0f} {ms/reps:9.3f} {100*ms/tot:6.1f} {flop/ms/1e9:8.1f}")

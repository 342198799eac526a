"""Quick A2J throughput probe (development aid; bench.py is the contract)."""
import sys, time
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = A2JEngine(synth.make_a2j_state_dict(0), device="cuda", precision=prec)
x = synth.make_crops(bs, 176, 3000).cuda()
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
t = time.time()
for _ in range(steps):
    out = eng.forward(x)
torch.cuda.synchronize()
dt = (time.time() - t) / steps
gf = 2 * eng.macs_per_crop() * bs / 1e9
print(f"A2J {prec} bs={bs}: {dt*1e3:.2f} ms/step  {bs/dt:.1f} crops/s  {gf/dt/1e3:.1f} TFLOP/s (algorithmic {gf/bs:.3f} GFLOP/crop)")

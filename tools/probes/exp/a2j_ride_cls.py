"""Does the classification head riding on layer4's launches (A2JEngine._trunk_multi, taken for <= 4 crops) pay at 32 / 64 crops?
usage (GPU box): python tools/probes/exp/a2j_ride_cls.py"""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch  # noqa: E402
from hn_amd import ops, synth  # noqa: E402
from hn_amd.a2j_engine import A2JEngine  # noqa: E402

sd = synth.make_a2j_state_dict(0)
for batch in (8, 32, 64):
    x = synth.make_crops(batch, 176, seed=3000).cuda()
    res = {}
    for limit in (4, 64):
        A2JEngine.MULTI_MAX_CROPS = limit
        eng = A2JEngine(sd, device="cuda")
        ref = eng.forward(x)
        with torch.no_grad(), ops.launch_cost_hidden():
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2): eng.forward(x)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): out = eng.forward(x)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): g.replay()
        b.record(); torch.cuda.synchronize()
        res[limit] = (a.elapsed_time(b) / 50, out.clone())
    print(f"batch {batch}: grouped heads {res[4][0]:.3f} ms   cls head riding on layer4 {res[64][0]:.3f} ms   max |d kp| {(res[4][1] - res[64][1]).abs().max().item():.2e}")

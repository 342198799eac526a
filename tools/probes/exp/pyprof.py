"""Where does the host time of an eager batch-1 step go?  (development aid)"""
import cProfile, pstats, sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd.fcos_engine import FCOSEngine
from hn_amd.pipeline import HandNetEngine
eng = HandNetEngine(FCOSEngine(synth.make_fcos_state_dict(0, 3), 3), A2JEngine(synth.make_a2j_state_dict(0)), 3)
rgb, depth = synth.make_rgb(1).cuda(), synth.make_depth(1).cuda()
for _ in range(5):
    eng.forward_device(rgb, depth)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    eng.forward_device(rgb, depth)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)

"""Where do the ~80 us go that HandNet.forward costs beyond the engine's replay at batch 1?  cProfile over 300 calls of the
drop-in at batch 1 (after it has switched itself to hipGraph replay), top functions by own time; plus the wall time of the
call split into "until the launch returns", "the sync", "after".   usage (GPU box): python tools/diag/dropin_host_profile.py"""
import cProfile
import pstats
import sys
import time
import types
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
from handnet_pipeline.handnet_pipeline import HandNet  # noqa: E402
from hn_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
fcos_sd, a2j_sd = synth.make_fcos_state_dict(seed=0, num_classes=3), synth.make_a2j_state_dict(seed=0)
net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
net.detector.load_state_dict(fcos_sd, strict=False)
net.a2j.load_state_dict(a2j_sd, strict=False)
net = net.to(dev).eval()
rgb, depth = synth.make_rgb(1, seed=1000).to(dev), synth.make_depth(1, seed=2000).to(dev)
images = [rgb[0]]
with torch.inference_mode():
    for _ in range(20):
        net(images, depth_images=depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        net(images, depth_images=depth)
    torch.cuda.synchronize()
    print(f"per call {1e6 * (time.perf_counter() - t0) / 300:.1f} us")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        net(images, depth_images=depth)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)

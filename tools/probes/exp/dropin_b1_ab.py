"""Same-box A/B of the drop-in's device -> host copy at batch 1: ops.to_host (pinned staging) vs Tensor.cpu()."""
import sys, time, types
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops, synth
from handnet_pipeline.handnet_pipeline import HandNet

args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j="unused.pth")
net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=False, RGBD=False)
net.detector.load_state_dict(synth.make_fcos_state_dict(0, 3), strict=False)
net.a2j.load_state_dict(synth.make_a2j_state_dict(0), strict=False)
net = net.cuda().eval()
rgb = [synth.make_rgb(1, seed=1000)[0].cuda()]
depth = synth.make_depth(1, seed=2000).cuda()
real = ops.to_host
with torch.inference_mode():
    for _ in range(30):
        net(rgb, depth_images=depth)
    for rep in range(3):
        for name, fn in (("Tensor.cpu()", lambda t: t.cpu()), ("ops.to_host", real)):
            ops.to_host = fn
            for _ in range(20):
                net(rgb, depth_images=depth)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                net(rgb, depth_images=depth)
            torch.cuda.synchronize()
            print(f"{name:14s} {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per call", flush=True)

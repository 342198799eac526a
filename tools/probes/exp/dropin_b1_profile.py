"""Where a batch-1 HandNet.forward call spends its host time (cProfile) and what the GPU runs outside the captured graph."""
import cProfile, pstats, sys, time, types
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import synth
from handnet_pipeline.handnet_pipeline import HandNet

args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j="unused.pth")
net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=False, RGBD=False)
net.detector.load_state_dict(synth.make_fcos_state_dict(0, 3), strict=False)
net.a2j.load_state_dict(synth.make_a2j_state_dict(0), strict=False)
net = net.cuda().eval()
rgb = [synth.make_rgb(1, seed=1000)[0].cuda()]
depth = synth.make_depth(1, seed=2000).cuda()
with torch.inference_mode():
    for _ in range(30):
        net(rgb, depth_images=depth)
    torch.cuda.synchronize()
    # graph replay alone
    eng = net.engine()
    key = next(iter(eng._graphs))
    g = eng._graphs[key][0]
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph replay alone        {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
        torch.cuda.current_stream().synchronize()
    print(f"graph replay + sync each  {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(300):
        net(rgb, depth_images=depth)
    torch.cuda.synchronize()
    print(f"HandNet.forward           {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        net(rgb, depth_images=depth)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)

"""Soak of the whole pipeline for run-to-run differences (a race in a hand-synchronised kernel -- LDS rings, counted waits, the
split-K tickets, chunked compaction -- shows as one): N steps of the batch-32 pipeline and of the single-frame pipeline, eager
and under hipGraph replay, every output compared bit for bit with the first step's ON THE DEVICE (one host read at the end).
usage (GPU box): python tools/soak.py [steps_b32] [steps_b1] [steps_live]"""
import sys
import time
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
from hn_amd import synth  # noqa: E402
from hn_amd.a2j_engine import A2JEngine  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402
from hn_amd.pipeline import HandNetEngine  # noqa: E402

steps32 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps1 = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
fcos_sd, a2j_sd = synth.make_fcos_state_dict(seed=0, num_classes=3), synth.make_a2j_state_dict(seed=0)
eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
total_bad = 0
for n, steps in ((32, steps32), (1, steps1)):
    rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
    ref = eng.forward_device(rgb, depth)
    kp0, box0, has0 = ref.keypoints.clone(), ref.crop_box.clone(), ref.has_hand.clone()
    for mode in ("eager", "graph"):
        bad = torch.zeros((), device="cuda", dtype=torch.int64)
        t0 = time.time()
        if mode == "graph":
            run, s_img, s_dep, out = eng.graphed(rgb, depth)
            s_img.copy_(rgb)
            s_dep.copy_(depth)
        k = steps if mode == "graph" else max(steps // 10, 50)
        for i in range(k):
            if mode == "graph":
                run()
            else:
                out = eng.forward_device(rgb, depth)
            bad += (out.keypoints.view(torch.int32) != kp0.view(torch.int32)).sum() + (out.crop_box != box0).sum() + \
                (out.has_hand != has0).sum()
            if i % 2000 == 1999:
                print(f"  batch {n} {mode}: step {i + 1}", flush=True)
        b = int(bad)
        total_bad += b
        print(f"batch {n} {mode}: {k} steps, {b} differing output words, {time.time() - t0:.1f} s", flush=True)
# round 6: the live chain (HandNet + aggregation-epilogue conversion + lifter: fused graph convolutions, matrix-vector PoseNet)
# as one captured step on one frame, every device output of every replay compared with the first
steps_live = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
if steps_live:
    import numpy as np
    import scipy.sparse as sp
    from hn_amd.live import LiveHandEngine
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    g = np.load(R / "tests" / "golden" / "pose2mesh_forward.npz")
    graphs = [sp.csr_matrix((g[f"L{i}_data"], g[f"L{i}_indices"], g[f"L{i}_indptr"]), shape=tuple(int(v) for v in g[f"L{i}_shape"]))
              for i in range(int(g["num_levels"]))]
    lifter = Pose2MeshEngine(synth.make_pose2mesh_state_dict(0, graph_sizes=[m.shape[0] for m in graphs]), graphs, device="cuda")
    live = LiveHandEngine(eng, lifter, (617.343, 617.343, 312.42, 241.42), clamp=True)
    rgb, depth = synth.make_rgb(1, seed=1000).cuda(), synth.make_depth(1, seed=2000).cuda()
    run, s_img, s_dep, out = live.graphed(rgb, depth)
    run()
    torch.cuda.synchronize()
    first = [t.clone() for t in (out.hand.keypoints, out.hand.image_uvd, out.hand.xyz_mm, out.mesh, out.pose3d)]
    bad = torch.zeros((), device="cuda", dtype=torch.int64)
    t0 = time.time()
    for i in range(steps_live):
        run()
        for a, b in zip((out.hand.keypoints, out.hand.image_uvd, out.hand.xyz_mm, out.mesh, out.pose3d), first):
            bad += (a.contiguous().view(torch.int32) != b.contiguous().view(torch.int32)).sum()
        if i % 5000 == 4999:
            print(f"  live batch 1 graph: step {i + 1}", flush=True)
    b = int(bad)
    total_bad += b
    print(f"live batch 1 graph: {steps_live} steps, {b} differing output words, {time.time() - t0:.1f} s", flush=True)
    from hn_amd import ops
    print("split-K tickets not at rest:", ops.tickets_nonzero())
print("TOTAL differing output words:", total_bad)
sys.exit(1 if total_bad else 0)

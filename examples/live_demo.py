#!/usr/bin/env python3
"""The reference's live loop (ros_demo.py:260-337) on the MI355X path, without ROS: synthetic 640x480 RGB-D frames go through

  1. the drop-in callable exactly as ros_demo.py:270 calls it            -> (keypoints on the CPU, depth crop, crop box)
  2. the same call with the caller's convert_joints folded into the step   -> net.last_converted (image uv, camera xyz in mm)
  3. the whole chain as ONE captured step (network -> convert -> Pose2Mesh lifter -> one copy)  -> mesh vertices

Weights are the seeded synthetic checkpoints of the tests (the published models/*.pth are not redistributable); with real
checkpoints pass their paths in `args` and reload_detector / reload_a2j = True, as ros_demo.py:370-388 does.
usage (GPU box): python examples/live_demo.py [frames]"""
import sys
import time
import types
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))                       # the drop-in tree in front of a reference checkout
sys.path.insert(0, str(REPO / "handnet-pipeline_amd" / "pose2mesh" / "lib"))

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

import models  # noqa: E402  (pose2mesh/lib/models, ros_demo.py:30)
from handnet_pipeline.handnet_pipeline import HandNet  # noqa: E402  (ros_demo.py:10)
from hn_amd import synth  # noqa: E402

PARAS = (617.343, 617.343, 312.42, 241.42)        # fx, fy, cx, cy of the depth camera (ros_demo.py:191-196)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    args = types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-")
    net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=False)           # ros_demo.py:374-388
    net.detector.load_state_dict(synth.make_fcos_state_dict(0, 3), strict=False)
    net.a2j.load_state_dict(synth.make_a2j_state_dict(0), strict=False)
    net = net.cuda().eval()
    g = np.load(REPO / "tests" / "golden" / "pose2mesh_forward.npz")                      # graph hierarchy (data fixture)
    graph_L = [sp.csr_matrix((g[f"L{i}_data"], g[f"L{i}_indices"], g[f"L{i}_indptr"]), shape=tuple(int(v) for v in g[f"L{i}_shape"]))
               for i in range(int(g["num_levels"]))]
    model = models.pose2mesh_net.get_model(21, graph_L)                                  # ros_demo.py:142
    model.load_state_dict(synth.make_pose2mesh_state_dict(0, graph_sizes=[m.shape[0] for m in graph_L]), strict=False)
    model = model.cuda().eval()

    rgb, depth = synth.make_rgb(1, seed=1000).cuda(), synth.make_depth(1, seed=2000).cuda()
    with torch.inference_mode():
        # 1. ros_demo.py:270
        keypoint_pred, depth_im, detections = net([rgb[0]], depth_images=depth)
        print("1. HandNet.forward:", tuple(keypoint_pred.shape), keypoint_pred.device, tuple(depth_im.shape), detections[0].tolist())
        # 2. the caller's clamp + convert_joints (ros_demo.py:279-290,329-330) as part of the step
        net.set_convert(PARAS, clamp=True)
        net([rgb[0]], depth_images=depth)
        conv = net.last_converted
        print("2. set_convert: joints2d[0] =", conv["image_uvd"][0, 0, :2].tolist(), " joints3d[0] (mm) =", conv["xyz_mm"][0, 0].tolist())
        # 3. the live chain as one captured step
        rev = torch.from_numpy(g["perm_reverse"][:778].astype(np.int64))                  # graph_perm_reverse[:V], ros_demo.py:162
        live = net.live(model, PARAS, clamp=True, perm_reverse=rev)                      # -> the step hands over out['mesh']
        run, s_img, s_dep, out = live.graphed(rgb, depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            s_img.copy_(synth.make_rgb(1, seed=1000 + i).cuda())                           # (a camera would write here)
            s_dep.copy_(synth.make_depth(1, seed=2000 + i).cuda())
            run()
            torch.cuda.current_stream().synchronize()
            kp, has_hand, crop_box, _words, (image_uvd, xyz_mm), mesh = out.read()
        dt = time.perf_counter() - t0
        cam_mesh = mesh[0]                                                                # out['mesh'] of ros_demo.py:332-337
        print(f"3. live step: {frames} frames, {1e3 * dt / frames:.2f} ms per frame incl. synthetic frame generation; "
              f"mesh {tuple(out.raw_mesh.shape)} -> {tuple(cam_mesh.shape)} camera-frame vertices; has_hand = {has_hand.tolist()}")


if __name__ == "__main__":
    main()

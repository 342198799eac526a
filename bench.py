#!/usr/bin/env python3
"""Throughput benchmark of the FCOS -> crop -> A2J hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms run N ranks, one per GPU.  For N > 1 the process that is started (by hand, or by torchrun as one of its ranks)
never touches the GPU: it is the SUPERVISOR of fresh worker processes (one per rank it owns) and runs the launch ladder --
if the workers die BEFORE the first timed step because the process group / RCCL cannot start (the IPC-handle mode is a host
property this file cannot know in advance), it starts ONE fresh set with HSA_ENABLE_IPC_MODE_LEGACY toggled (default "0" ->
unset) and the line reports which mode ran (`config.ipc_mode`).  Never a gloo fallback, never a re-exec, never a silent
single-GPU run: a world size that differs from --gpus is an error, and so is a second failed attempt (the backend's error
text is printed).  Every attempt rendezvouses through its own file:// store with a bounded timeout (--init-timeout).

One "step" = one pass of HandNet (FCOS detector, top-1 hand crop, A2J) over one batch of
synthetic 640x480 RGB-D frames already resident in HBM, plus -- for N > 1 -- the all-gather
of the per-frame results (RCCL).  Weak scaling: every rank processes its own `--batch`
frames (BASELINE.json config 4: 32 frames on one GPU; config 5: 8 x 32).  Rank 0 prints ONE
JSON line; `value` is whole-job frames/s = N * batch * K / max-over-ranks(time).

For N > 1 the step is replayed from a captured hipGraph by default (N Python hosts share the box's cores; --eager turns it
off); every rank reports its device (`config.devices`: PCI bus ids, asserted distinct) and its own time per step
(`rank_ms`: [min, median, max] over the ranks of the HIP-event time of the engine's launches, so a straggler is visible
beside the max-over-ranks wall time that `value` is computed from).

Extra objects in the line:
  other_configs the other single-GPU configurations of BASELINE.json (default N = 1 run only): a2j_b64 (config 2), fcos_b16
                (config 3), pipeline_b1 (the reference caller's own batch, ros_demo.py:270) -- hipGraph replay, median of five
                groups of steps; value, ms_per_step and the dominant kernel's roofline fraction of each -- and pipeline_b32_f32:
                config 4 on the exact-f32 engines (the reference's own arithmetic; 3 timed steps, fraction of the f32-MFMA
                peak); pose2mesh_b1 / _b32: the lifter that follows the path alone; live_b1: HandNet + convert_joints
                + lifter as ONE captured step with one device -> host copy (the live caller's chain, ros_demo.py:270-337).  a2j_b64.parity ("EPE vs CPU ref": max |d(u,v,d)| and mm EPE of 16 of the 64 crops vs the oracle) and
                fcos_b16.parity ("box IoU vs torchvision ref": survivor-index equality, labels, mean / min IoU of matched
                survivors on the 16 frames) compare the outputs of the TIMED steps; the oracle runs after every timed region.
  roofline      dominant kernel (the conv_igemm_f16x3_kernel instantiation with the largest share of
                the step; conv_igemm_f32_kernel with --precision f32): ALGORITHMIC FLOP per launch /
                average launch duration, both measured live with HIP events on the launch stream
                over instrumented steps, vs the dense f16 MFMA peak of 2.5 PFLOP/s (157.3 TFLOP/s
                f32-MFMA peak in f32 mode; MI355X_MICROARCH.md).  f16x3 issues 3 MFMAs per
                algorithmic MAC; the issued rate is reported beside it.  `clock_mhz` is the shader
                clock sampled in-kernel (s_memtime / s_memrealtime) on a side stream WHILE the
                instrumented steps run.  `traffic` is the HBM byte count per launch from the committed
                rocprofv3 PMC passes (profiles/traffic_latest.json) and is null unless that
                collection was made at the same launches/step and FLOP/launch (`traffic_source`).
                `stages` splits the instrumented steps by stage of the hot path (ResNet-34 body / FPN / towers /
                head outputs / A2J trunk / A2J heads): ms per step, algorithmic TFLOP/s and fraction of the peak.
  dropin        the same batch through the reference's own callable -- handnet_pipeline.HandNet.forward as
                ros_demo.py:270-273 calls it (list of [3,H,W] images + depth_images, keypoints copied to the
                CPU) -- at the bench batch and at batch 1 (the only batch size the reference's caller uses); and
                `batch*_host`: the PCIe-inclusive figures, the same calls with the frames starting in pinned host
                memory -- as converted fp32 tensors, and as the raw bgr8 + 16UC1 buffers through HandNet.forward_raw.
  cpu_baseline  the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample of the same
                workload (rank 0, N == 1 only): ONE pass over --cpu-frames frames in batches of 8 (`value`), and `batch1`: the
                median of 5 single-frame forwards after 2 warm-ups; `sample` says so.  The same frames go through the HIP
                engine and the agreement is reported as `parity`.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

# the host driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames (or crops) per GPU; default 32 (a2j: 64)")
    ap.add_argument("--workload", choices=["pipeline", "a2j", "fcos", "pose2mesh", "live"], default="pipeline",
                    help="live: the live caller's chain (HandNet + convert_joints + Pose2Mesh lifter, one copy) as one step")
    ap.add_argument("--precision", choices=["f16x3", "f32", "f16x1"], default="f16x3",
                    help="f16x3: split-fp16 operands on the f16 MFMA (fp32-grade results; the headline); f32: exact f32 MFMA; "
                         "f16x1: hi*hi term only = plain fp16 operands, 1 MFMA per MAC -- the THROUGHPUT mode SURVEY D6 plans "
                         "beside the parity mode: misses the 1e-3 contract, reported with its error figures, never the headline")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (the default for --gpus > 1)")
    ap.add_argument("--eager", action="store_true", help="--gpus > 1: issue the step's launches from Python instead of replaying a graph")
    ap.add_argument("--capture-gather", action="store_true",
                    help="--gpus > 1: ALSO capture the all-gather into the step's hipGraph (ShardedHandNet's own default; verified on a "
                         "single-rank RCCL group, never yet on a multi-rank one: the default here keeps the gather eager behind the "
                         "replayed step, the form rehearsed since round 3)")
    ap.add_argument("--init-timeout", type=float, default=180.0,
                    help="--gpus > 1: seconds the process group may take to start (rendezvous + first collective) per attempt")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other single-GPU BASELINE configurations (a2j_b64, fcos_b16, pipeline_b1) of the default run")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST ONLY: a CPU stand-in for the engine (sleeps, deterministic records) so that the N-rank plumbing of "
                         "this file -- ladder, rendezvous, gather, line -- runs on gloo without a GPU; the line says so")
    ap.add_argument("--native", action="store_true",
                    help="drive the step through the model-level C ABI (C++ layer graphs, csrc/model.hip) instead of "
                         "the Python engines; same launches, same results")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the second figure: the same batch through the reference's callable HandNet.forward (+ batch 1)")
    ap.add_argument("--no-clock-sample", action="store_true",
                    help="skip the one-wave in-kernel clock sampler of the roofline leg (it runs beside the step on a side "
                         "stream and would show up as a long kernel in rocprofv3 --stats)")
    ap.add_argument("--cpu-frames", type=int, default=32,
                    help="frames in the bounded CPU sample (also the frames of the HIP-vs-oracle parity figures)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N > 1 plumbing on a box with fewer GPUs than ranks "
                         "(results are gathered through host memory; not a performance mode)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def build_workload(args, dev, rank):
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine

    wl = args.workload
    batch = args.batch or (64 if wl == "a2j" else 32 if wl in ("pipeline", "pose2mesh") else 1 if wl == "live" else 16)
    if wl == "pose2mesh":   # the lifter that follows the path in the live demo
        import numpy as np
        import scipy.sparse as sp
        from hn_amd.pose2mesh_engine import Pose2MeshEngine
        g = np.load(REPO / "tests" / "golden" / "pose2mesh_forward.npz")   # graph hierarchy (data fixture)
        graphs = [sp.csr_matrix((g[f"L{i}_data"], g[f"L{i}_indices"], g[f"L{i}_indptr"]),
                                shape=tuple(int(v) for v in g[f"L{i}_shape"])) for i in range(int(g["num_levels"]))]
        sd = synth.make_pose2mesh_state_dict(0, graph_sizes=[m.shape[0] for m in graphs])
        eng = Pose2MeshEngine(sd, graphs, device=dev)
        x = torch.randn((batch, 21, 2), generator=torch.Generator().manual_seed(5000 + rank)).to(dev)
        macs = sum(int(v.numel()) for k, v in sd.items() if k.startswith("pose_lifter") and k.endswith("weight") and v.dim() == 2)
        macs += int(sd["pose2mesh.fc.weight"].numel())
        lv = [m.shape[0] for m in graphs]
        del lv[-2]
        from hn_amd.pose2mesh_engine import CL_F
        idx = 0
        for bi, chain in enumerate(CL_F):
            v = lv[-(bi + 1) + (1 if bi == len(CL_F) - 1 else 0)]
            for li in range(len(chain) - 1):
                macs += v * int(sd[f"pose2mesh.cl.{idx}.weight"].numel())
                idx += 1
        info = {"batch_per_gpu": batch, "unit": "meshes/s", "gflop_per_unit": 2 * macs / 1e9,
                "name": "Pose2Mesh lifter (PoseNet MLP + Chebyshev graph-conv mesh net), 21 joints -> 1152-vertex hierarchy"}
        if args.graph:
            run, _, out = eng.graphed(x)

            def step():
                run()
                return out
            return step, info, None
        return (lambda: eng.forward(x)), info, None
    a2j_sd = synth.make_a2j_state_dict(0)
    fcos_sd = synth.make_fcos_state_dict(0, 3)
    info = {"batch_per_gpu": batch}
    if wl == "live":   # ros_demo.py:270-290,329-337 as one step (hn_amd/live.py); profiling workload of tools/collect_profiles.sh
        import numpy as np
        import scipy.sparse as sp
        from hn_amd.live import LiveHandEngine
        from hn_amd.pose2mesh_engine import Pose2MeshEngine
        g = np.load(REPO / "tests" / "golden" / "pose2mesh_forward.npz")
        graphs = [sp.csr_matrix((g[f"L{i}_data"], g[f"L{i}_indices"], g[f"L{i}_indptr"]),
                                shape=tuple(int(v) for v in g[f"L{i}_shape"])) for i in range(int(g["num_levels"]))]
        lifter = Pose2MeshEngine(synth.make_pose2mesh_state_dict(0, graph_sizes=[m.shape[0] for m in graphs]), graphs, device=dev)
        hand = HandNetEngine(FCOSEngine(fcos_sd, 3, device=dev, precision=args.precision),
                             A2JEngine(a2j_sd, device=dev, precision=args.precision), 3)
        live = LiveHandEngine(hand, lifter, LIVE_PARAS, clamp=True)
        rgb, depth = synth.make_rgb(batch, seed=1000 + rank).to(dev), synth.make_depth(batch, seed=2000 + rank).to(dev)
        info.update(unit="frames/s", gflop_per_unit=2 * (hand.fcos.macs_per_frame() + hand.a2j.macs_per_crop()) / 1e9,
                    name="the live caller's chain: HandNet -> convert_joints -> Pose2Mesh lifter -> one copy (ros_demo.py:270-337)")
        if args.graph:
            run, _, _, out = live.graphed(rgb, depth)

            def step():
                run()
                return out.hand
            return step, info, None
        return (lambda: live.forward_device(rgb, depth).hand), info, None
    if wl == "a2j":
        eng = A2JEngine(a2j_sd, device=dev, precision=args.precision)
        x = synth.make_crops(batch, 176, seed=3000 + rank).to(dev)
        step = lambda: eng.forward(x)  # noqa: E731
        info.update(unit="crops/s", gflop_per_unit=2 * eng.macs_per_crop() / 1e9,
                    name="A2J-only inference, 176x176 depth crops (BASELINE config 2)")
        return step, info, None
    fcos = FCOSEngine(fcos_sd, 3, device=dev, precision=args.precision)
    rgb = synth.make_rgb(batch, seed=1000 + rank).to(dev)
    if wl == "fcos":
        step = lambda: fcos.detect(rgb)  # noqa: E731
        info.update(unit="frames/s", gflop_per_unit=2 * fcos.macs_per_frame() / 1e9,
                    name="FCOS ResNet34-FPN detector, 640x480 RGB (BASELINE config 3)")
        return step, info, None
    a2j = A2JEngine(a2j_sd, device=dev, precision=args.precision)
    depth = synth.make_depth(batch, seed=2000 + rank).to(dev)
    eng = HandNetEngine(fcos, a2j, 3)
    info.update(unit="frames/s", gflop_per_unit=2 * (fcos.macs_per_frame() + a2j.macs_per_crop()) / 1e9,
                name="Full HandNet pipeline (FCOS -> crop -> A2J), 640x480 RGB-D (BASELINE config 4)", engine=eng)
    info["eager_step"] = lambda: eng.forward_device(rgb, depth)  # noqa: E731 -- (the roofline leg instruments eager launches)
    info["inputs"] = (rgb, depth)
    if args.native:
        from hn_amd.native_model import NativeModel
        from hn_amd.pipeline import HandNetOutput
        native = NativeModel(fcos_sd, a2j_sd, num_classes=3, device=dev, precision=args.precision)
        info["native"] = native

        def step():
            kp, box, has = native.handnet(rgb, depth)
            return HandNetOutput(kp, None, box, has, None, None)
    elif args.graph:
        run, _, _, out = eng.graphed(rgb, depth)

        def step():
            run()
            return out
    else:
        step = lambda: eng.forward_device(rgb, depth)  # noqa: E731
    return step, info, (fcos_sd, a2j_sd)


def roofline_leg(step, steps, ms_per_step, sample_clock=True, terms=3):
    """Bracket every conv launch with HIP events (on the launch stream) for `steps` steps; a one-wave sampler on a
    side stream reads the shader clock the chip holds meanwhile."""
    from hn_amd import ops
    side = torch.cuda.Stream()
    clock = torch.zeros((1,), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    ops.CONV_PROFILE = []
    try:
        step()                                    # chip under load before the sampling window opens
        if sample_clock:
            ops.clock_sample(min(2_000_000, max(1000, int(0.8 * 1e3 * ms_per_step * steps))), out=clock, stream=side)
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        recs = ops.CONV_PROFILE
    finally:
        ops.CONV_PROFILE = None
    clock_mhz = float(clock.item())
    groups, stages = {}, {}
    for kind, macs, timer, shape, stage in recs:
        g = groups.setdefault(kind, {"ms": 0.0, "flop": 0.0, "launches": 0, "bytes": 0.0})
        ms = timer.elapsed_ms()
        g["ms"] += ms
        g["flop"] += 2.0 * macs
        g["launches"] += 1
        st = stages.setdefault(stage or "other", {"ms": 0.0, "flop": 0.0, "launches": 0})
        st["ms"] += ms
        st["flop"] += 2.0 * macs
        st["launches"] += 1
        n, h, w, cin, cout, r, stride, _dil = shape
        # algorithmic bytes: input + filters + output, 4 bytes per value (fp32 or S32 hi+lo)
        g["bytes"] += 4.0 * (n * h * w * cin + cout * r * r * cin + n * (h // stride) * (w // stride) * cout)
    if not groups:
        return None
    (prec, tile), g = max(groups.items(), key=lambda kv: kv[1]["ms"])
    tot_ms = sum(v["ms"] for v in groups.values())
    tot_flop = sum(v["flop"] for v in groups.values())
    steps += 1                                    # the load-up step was instrumented too
    achieved = g["flop"] / (g["ms"] * 1e-3) / 1e12
    # f16x3 issues 3 f16 MFMAs per algorithmic MAC; `achieved` stays ALGORITHMIC FLOP/s and is
    # priced against the dense f16 MFMA peak (the issued-MFMA rate is reported next to it).
    peak = F16_MFMA_PEAK_TFLOPS if prec == "f16x3" else F32_MFMA_PEAK_TFLOPS
    kernel = f"conv_igemm_{prec}_kernel<{ops.tile_name(tile)}>"
    gf_launch = round(g["flop"] / g["launches"] / 1e9, 3)
    traffic, traffic_source = measured_traffic(kernel, gf_launch, g["launches"] // steps)
    issued = achieved * (terms if prec == "f16x3" else 1)
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": int(g["bytes"] / g["launches"]),
        "kernel": kernel,
        # shader clock sampled in-kernel while the instrumented steps ran, and the same fraction against the MFMA
        # peak AT that clock (the nominal peak assumes 2400 MHz; under this load the chip holds less)
        "clock_mhz": round(clock_mhz, 1) if clock_mhz > 0 else None,
        "frac_at_clock": round(achieved / (peak * clock_mhz / 2400.0), 4) if clock_mhz > 0 else None,
        "mfma_issued_frac_at_clock": round(issued / (peak * clock_mhz / 2400.0), 4) if clock_mhz > 0 else None,
        "mfma_issued_tflops": round(issued, 2),
        "mfma_issued_frac": round(issued / peak, 4),
        "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
        "gflop_per_launch": gf_launch,
        "launches_per_step": g["launches"] // steps,
        "share_of_conv_time": round(g["ms"] / tot_ms, 3),
        "all_conv_achieved": round(tot_flop / (tot_ms * 1e-3) / 1e12, 2),
        "conv_ms_per_step": round(tot_ms / steps, 3),
        # convolution launches of the instrumented steps by stage of the hot path (north_star: "fraction of the conv
        # roofline on the ResNet stages"): HIP-event time per step, algorithmic TFLOP/s and the fraction of `peak`
        "stages": {name: {"ms_per_step": round(v["ms"] / steps, 3), "launches_per_step": v["launches"] // steps,
                          "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1),
                          "frac": round(v["flop"] / (v["ms"] * 1e-3) / 1e12 / peak, 4)}
                   for name, v in stages.items() if v["ms"] > 0},
    }


def measured_traffic(kernel, gflop_per_launch, launches_per_step):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic_latest.json, written by tools/save_profiles.py).  The collection is stamped with the
    launches/step and GFLOP/launch of the dominant kernel it was made at; a figure collected for another
    configuration (an older commit, another batch) is NOT paired with this run: (None, reason)."""
    f = REPO / "profiles" / "traffic_latest.json"
    if not f.exists():
        return None, "no profiles/traffic_latest.json"
    rec = json.loads(f.read_text())
    stamp = rec.get("bench")
    if not stamp:
        return None, f"{f.name} ({rec.get('tag')}) carries no bench stamp"
    if rec.get("kernel_source_sha16") != kernel_source_sha16():
        return None, (f"{rec.get('tag')} (commit {rec.get('commit')}): collected for another revision of "
                      "csrc/conv_igemm_f16x3_kernel.h (data movement may differ)")
    if stamp.get("kernel") != kernel:
        return None, f"{rec.get('tag')}: collected for {stamp.get('kernel')}, this run's dominant kernel is {kernel}"
    if (stamp.get("launches_per_step") != launches_per_step
            or abs(stamp.get("gflop_per_launch", 0.0) - gflop_per_launch) > 5e-3 * gflop_per_launch):
        return None, (f"{rec.get('tag')} (commit {rec.get('commit')}): collected at {stamp.get('launches_per_step')} "
                      f"launches/step x {stamp.get('gflop_per_launch')} GFLOP, this run has {launches_per_step} x "
                      f"{gflop_per_launch}")
    want = stamp["kernel"].replace("conv_igemm_", "").split("_kernel<")
    prec, tile = want[0], want[1].rstrip(">")
    rs = tile.endswith("+rs")                      # the row-shared-A instantiation
    bm, bn = tile.replace("+rs", "").split("x")
    needle = f"conv_igemm_{prec}_kernel<{bm}, {bn},"
    for name, k in rec["kernels"].items():
        if needle not in name:
            continue
        # template arguments <BM, BN, WM, WN, NBUF, BUF, RS, TERMS> (f16x3; rounds 1-4 had a POOL flag in front of TERMS) /
        # <BM, BN, SMALLC> (f32)
        targs = [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")]
        if prec == "f16x3":
            if len(targs) == 9 and targs[7] == "false":
                del targs[7]
            if len(targs) != 8 or (targs[6] == "true") != rs or targs[7] != "3":
                continue
        return k["hbm_bytes_per_launch"], f"profiles/{rec.get('tag')}_traffic.json (commit {rec.get('commit')})"
    return None, f"{rec.get('tag')}: no {needle} record"


def kernel_source_sha16():
    """Identity of the dominant kernel's source: the HBM-traffic figure of an older revision is never paired with a run."""
    import hashlib
    return hashlib.sha256((REPO / "handnet-pipeline_amd" / "csrc" / "conv_igemm_f16x3_kernel.h").read_bytes()).hexdigest()[:16]


def dropin_leg(args, sds, dev, batch):
    """The SAME workload through the reference's callable: handnet_pipeline.HandNet.forward exactly as ros_demo.py:270-273
    calls it (a list of [3,H,W] images, depth_images=[N,1,H,W]; keypoints come back on the CPU, which is a device -> host
    copy and a sync per call), at the bench batch and at batch 1 (the reference's caller never batches)."""
    import types

    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    fcos_sd, a2j_sd = sds
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.to(dev).eval()
    out = {"call": "handnet_pipeline.HandNet.forward(list of [3,H,W], depth_images=[N,1,H,W]) -> (keypoints on the CPU, "
                   "depth_batch, crops), ros_demo.py:270-273",
           # the precision the drop-in's engines REALLY run at (its constructor has no precision argument: always the
           # parity-grade default, whatever --precision the engine-level legs of this run use)
           "precision": "f16x1" if net.engine().fcos.terms == 1 else net.engine().fcos.precision,
           "host": "forward() has switched itself to hipGraph replay (its default once the input shapes repeat on a dense "
                   "stream; results are fresh tensors either way)"}
    for b, steps in ((batch, args.steps), (1, max(50, args.steps))):
        rgb = synth.make_rgb(b, seed=1000).to(dev)
        depth = synth.make_depth(b, seed=2000).to(dev)
        images = [rgb[i] for i in range(b)]
        with torch.inference_mode():
            for _ in range(6):   # (a batch below 8 switches itself to hipGraph replay at the fifth same-shape call)
                net(images, depth_images=depth)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                kp, _db, _crops = net(images, depth_images=depth)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert kp.device.type == "cpu" and tuple(kp.shape) == (b, 21, 3)
        out[f"batch{b}"] = {"frames_per_s": round(b * steps / dt, 2), "ms_per_call": round(1e3 * dt / steps, 3), "calls": steps}
        # PCIe-INCLUSIVE: the same call when the frames start in (pinned) HOST memory, as they do for the reference's caller
        # (ros_demo.py:227-231,266-267).  fp32 feed: the host has already converted (4.9 MB per frame cross PCIe as
        # `.cuda()` copies); raw feed: HandNet.forward_raw on the cv_bridge 'bgr8' + 16UC1 buffers (1.5 MB per frame, the
        # conversion runs in the ingest kernel, which reads the pinned buffers itself).
        import numpy as np
        rng = np.random.default_rng(1000)
        bgr = torch.from_numpy(rng.integers(0, 256, size=(b, 480, 640, 3), dtype=np.uint8)).pin_memory()
        mm = torch.from_numpy(rng.integers(300, 1500, size=(b, 480, 640)).astype(np.uint16)).pin_memory()
        rgb_h = (bgr.flip(-1).permute(0, 3, 1, 2).float() / 255.0).contiguous().pin_memory()
        dep_h = (mm.float() / 1000.0).unsqueeze(1).contiguous().pin_memory()
        legs = {}
        with torch.inference_mode():
            def fp32_feed():
                return net([rgb_h[i].to(dev, non_blocking=True) for i in range(b)], depth_images=dep_h.to(dev, non_blocking=True))

            def raw_feed():
                return net.forward_raw(bgr, mm)
            for name, call in (("fp32_feed", fp32_feed), ("raw_feed", raw_feed)):
                for _ in range(6):
                    call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    kp, _db, _crops = call()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                legs[name] = {"frames_per_s": round(b * steps / dt, 2), "ms_per_call": round(1e3 * dt / steps, 3)}
        legs["bytes_over_pcie_per_frame"] = {"fp32_feed": 4 * 480 * 640 * 4, "raw_feed": 480 * 640 * 5}
        legs["what"] = ("the call of batch%d with its inputs starting in pinned host memory: host -> device transfer inside the "
                        "timed region (value of the line: inputs resident in HBM)" % b)
        out[f"batch{b}_host"] = legs
    del net
    return out


def cpu_baseline(args, sds, engine=None, dev=None, others=None, hip_outputs=None):
    """Oracle (CPU restatement of the reference path) on a bounded sample, all host cores.  For the pipeline the
    same frames also go through the HIP engine, and the difference is reported next to the timing ("parity")."""
    from hn_amd import synth
    from oracle import a2j_ref, handnet_ref
    # one GPU's share of the host is 16 cores on the benchmark boxes; more threads than that
    # makes torch's CPU convolutions slower, not faster (measured: 256 threads = 50x slower)
    threads = int(os.environ.get("HN_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(threads)
    n = args.cpu_frames
    if args.workload == "a2j":
        x = synth.make_crops(16, 176, seed=3000)
        sd = synth.make_a2j_state_dict(0)
        a2j_ref.a2j_forward(x[:2], sd)
        t0 = time.time()
        reps = 3
        for _ in range(reps):
            a2j_ref.a2j_forward(x, sd)
        dt = time.time() - t0
        return {"value": round(16 * reps / dt, 2), "unit": "crops/s", "cores": threads, "kind": "port",
                "sample": f"{reps} x batch-16 A2J oracle forward (torch CPU fp32)"}
    fcos_sd, a2j_sd = sds
    rgb = synth.make_rgb(n, seed=1000)
    depth = synth.make_depth(n, seed=2000)
    if engine is None:
        imgs = [rgb[i] for i in range(n)]
        handnet_ref.handnet_forward(imgs[:1], depth[:1], fcos_sd, a2j_sd, 3)  # warm-up
        t0 = time.time()
        for lo in range(0, n, 8):
            handnet_ref.handnet_forward(imgs[lo:lo + 8], depth[lo:lo + 8], fcos_sd, a2j_sd, 3)
        dt = time.time() - t0
        return {"value": round(n / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                "sample": f"{n} frames in batches of 8, full-pipeline oracle forward (torch CPU fp32, FCOS+crop+A2J)"}
    # the same frames through the oracle (timed) and the HIP engine: where end-to-end parity can break
    from hn_amd import ops
    from oracle import parity
    handnet_ref.handnet_forward([rgb[0]], depth[:1], fcos_sd, a2j_sd, 3)  # warm-up
    stats, oracle_s, (g_kp, g_box, g_has, o_kp, o_box, o_has, o_dets) = parity.pipeline_parity(
        engine, rgb, depth, fcos_sd, a2j_sd, 3, chunk=8)
    res = {"value": round(n / oracle_s, 3), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"ONE pass over {n} frames (seeds 1000/2000) in batches of 8 after one warm-up frame: full-pipeline oracle "
                     f"forward (oracle.fcos_ref + handnet_ref.select_and_crop + a2j_ref, torch CPU fp32, {threads} threads); "
                     "the bounded form of BASELINE.md section 3 (a median of 5 passes would be 80 s of CPU); batch 1: `batch1`"}
    eq = (g_box == o_box).all(dim=1) & g_has & o_has
    if bool(eq.any()):
        paras = (617.343, 617.343, 312.42, 241.42)   # SURVEY 8d intrinsics for the millimetre figure
        valid = eq.to(torch.int32).to(dev)
        xyz = ops.convert_joints(g_kp.to(dev), g_box.to(dev), valid, paras).cpu()
        xyz_ref = ops.convert_joints(o_kp.to(dev), o_box.to(dev), valid, paras).cpu()
        stats["mm_epe"] = float((xyz - xyz_ref)[eq].norm(dim=-1).mean())
    res["parity"] = stats
    # batch 1, the reference caller's batch (ros_demo.py:270): 2 warm-ups, median of 5 single-frame oracle forwards
    per = []
    for i in range(7):
        t0 = time.time()
        handnet_ref.handnet_forward([rgb[i % n]], depth[i % n:i % n + 1], fcos_sd, a2j_sd, 3)
        per.append(time.time() - t0)
    med = sorted(per[2:])[2]
    res["batch1"] = {"value": round(1.0 / med, 3), "unit": "frames/s", "ms_per_frame": round(1e3 * med, 1),
                     "sample": "median of 5 single-frame full-pipeline oracle forwards after 2 warm-ups (frames 0..6 of the sample)"}
    # BASELINE configs 2 and 3 in their own parity terms, against what their TIMED steps produced (other_configs_leg)
    if others is not None and hip_outputs:
        if "fcos_b16" in hip_outputs:
            # frames 0..15 of the sample ARE the batch-16 detector run's frames (make_rgb streams one generator): the oracle's
            # detections for them come from the pass above (a sample of fewer than 16 frames compares what it has)
            m = min(16, n)
            others["fcos_b16"]["parity"] = parity.fcos_parity(hip_outputs["fcos_b16"][:m], o_dets[:m])
        if "a2j_b64" in hip_outputs:
            hip_kp, crops = hip_outputs["a2j_b64"]
            others["a2j_b64"]["parity"], a2j_s = parity.a2j_parity(hip_kp, crops, a2j_sd)
            res["a2j_only"] = {"value": round(crops.shape[0] / a2j_s, 2), "unit": "crops/s",
                               "sample": f"ONE batch-{crops.shape[0]} oracle.a2j_ref.a2j_forward (the parity run of other_configs.a2j_b64)"}
        if "live_b1" in hip_outputs:
            # frame 0 of the sample IS the live step's frame (seeds 1000/2000); the lifter's weights / graphs are the step's own
            hip_live, p2m_sd, graphs = hip_outputs["live_b1"]
            others["live_b1"]["parity"], live_s, lift_s = parity.live_parity(hip_live, rgb[:1], depth[:1], fcos_sd, a2j_sd, p2m_sd,
                                                                             graphs, LIVE_PARAS, 3)
            res["live_batch1"] = {"value": round(1.0 / live_s, 3), "unit": "frames/s", "ms_per_frame": round(1e3 * live_s, 1),
                                  "lifter_ms": round(1e3 * lift_s, 1),
                                  "sample": "median of 5 single-frame oracle chains after one warm-up: full-pipeline oracle forward + the "
                                            "caller's numpy glue + oracle.pose2mesh_ref (the parity run of other_configs.live_b1); "
                                            "lifter_ms = lifter_input + pose2mesh_forward alone"}
    return res


EXIT_INIT_FAILED = 75   # a worker's exit status when the process group could not start (before the first timed step)
IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"


def _ipc_label(env):
    return f"{IPC_VAR}={env[IPC_VAR]}" if IPC_VAR in env else f"{IPC_VAR} unset"


def _ladder_envs():
    """The two rungs of the launch ladder: the environment as it is (this file defaults the IPC variable to "0", which the
    one-GPU boxes need), then ONE alternative with the variable toggled ("0" -> unset; anything else -> "0")."""
    first = dict(os.environ)
    second = dict(os.environ)
    if first.get(IPC_VAR) == "0":
        second.pop(IPC_VAR)
    else:
        second[IPC_VAR] = "0"
    return [first, second]


def visible_gpus():
    """GPUs of this host without touching the HIP / amdsmi runtimes (torch.cuda.device_count() goes through amdsmi and falls
    back to hipGetDeviceCount when that fails -- the one process that must never initialise the GPU before it spawns its
    workers might): KFD topology nodes with a non-zero simd_count are GPUs (CPUs have 0), narrowed by an index list in
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology is not readable: the pre-check is then skipped (the
    workers' distinct-PCI-id assertion catches a short host anyway)."""
    nodes = Path("/sys/class/kfd/kfd/topology/nodes")
    try:
        count = 0
        for node in nodes.iterdir():
            props = dict(l.split(None, 1) for l in (node / "properties").read_text().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def supervise(args):
    """N > 1: this process never initialises HIP.  It owns the workers of one or more ranks (all N when started by hand, its
    own rank when torchrun started it), relays their output (inherited stdout / stderr) and walks the launch ladder: a worker
    that fails before it has written its `ready` marker -- i.e. before the process group has carried a collective -- marks the
    attempt as failed; every supervisor of the job sees the mark (shared directory), stops ITS workers by pid, and all start
    one fresh set on the next rung.  Returns the exit status."""
    import signal
    import tempfile
    under_torchrun = "WORLD_SIZE" in os.environ
    if under_torchrun:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as "
                             f"{args.gpus} GPUs")
        ranks = [(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")))]
        # one directory per job: the ranks of a torchrun job share their parent (the elastic agent)
        ppid = os.getppid()
        try:
            start = Path(f"/proc/{ppid}/stat").read_text().rsplit(")", 1)[1].split()[19]
        except Exception:  # noqa: BLE001
            start = "0"
        job_dir = Path(tempfile.gettempdir()) / f"hn_bench_{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}"
        job_dir.mkdir(exist_ok=True)
    else:
        if not args.stub_engine and not args.share_gpu:
            have = visible_gpus()                     # (sysfs: the supervisor never opens the GPU runtime)
            if have is not None and have < args.gpus:
                raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) are visible "
                                 "(--share-gpu --dist-backend gloo rehearses the plumbing on one)")
        world = args.gpus
        ranks = [(r, r) for r in range(world)]
        job_dir = Path(tempfile.mkdtemp(prefix="hn_bench_"))
    cmd = [sys.executable, str(Path(__file__).resolve())] + sys.argv[1:]
    children = []

    def stop_children(*_):
        for c in children:
            if c.poll() is None:
                c.terminate()
        for c in children:
            try:
                c.wait(timeout=15)
            except subprocess.TimeoutExpired:
                c.kill()
                c.wait()

    def on_signal(signum, _frame):
        stop_children()
        sys.exit(128 + signum)
    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)

    envs = _ladder_envs()
    status = 1
    for attempt, base in enumerate(envs):
        failed = job_dir / f"attempt{attempt}.failed"
        print(f"[bench] attempt {attempt + 1} of {len(envs)}: starting {len(ranks)} worker(s) of {world} ranks with "
              f"{_ipc_label(base)}", file=sys.stderr, flush=True)
        children.clear()
        for rank, local in ranks:
            env = dict(base, RANK=str(rank), LOCAL_RANK=str(local), WORLD_SIZE=str(world), HN_BENCH_WORKER="1",
                       HN_BENCH_ATTEMPT=str(attempt), HN_BENCH_DIR=str(job_dir), HN_BENCH_IPC_MODE=_ipc_label(base))
            env.setdefault("MASTER_ADDR", "127.0.0.1")
            children.append(subprocess.Popen(cmd, env=env))
        init_failed = False
        while True:
            codes = [c.poll() for c in children]
            for (rank, _), code in zip(ranks, codes):
                if code not in (None, 0) and not (job_dir / f"attempt{attempt}.rank{rank}.ready").exists():
                    init_failed = True        # died before the group carried a collective (whatever the status: a watchdog abort too)
                    if not failed.exists():
                        try:
                            failed.write_text(f"rank {rank}: worker exited with status {code} before the process group was up\n")
                        except OSError:
                            pass
            if failed.exists():
                init_failed = True
            # a worker that fails AFTER the group was up (a refused configuration, a crash in the run) ends the job: its peers
            # would otherwise wait in their next collective until the group's timeout
            run_failed = any(code not in (None, 0) and (job_dir / f"attempt{attempt}.rank{rank}.ready").exists()
                             for (rank, _), code in zip(ranks, codes))
            if init_failed or run_failed or all(c is not None for c in codes):
                if run_failed and not init_failed:
                    stop_children()
                break
            time.sleep(0.1)
        if init_failed:
            stop_children()
            why = failed.read_text().strip() if failed.exists() else "unknown"
            print(f"[bench] attempt {attempt + 1} failed to initialise: {why}", file=sys.stderr, flush=True)
            if attempt + 1 < len(envs):
                (job_dir / f"attempt{attempt + 1}.why").write_text(why[:400])
                continue
            status = EXIT_INIT_FAILED
            break
        status = max((c.returncode for c in children if c.returncode is not None), key=abs, default=1)
        if status == 0 and any(c.returncode != 0 for c in children):
            status = 1
        break
    if not under_torchrun or (status == 0 and any(r == 0 for r, _ in ranks)):
        import shutil
        if under_torchrun:
            time.sleep(1.0)    # (the other ranks' supervisors only look at the directory while their worker runs)
        shutil.rmtree(job_dir, ignore_errors=True)
    return status


class _StubOutput:
    def __init__(self, keypoints, crop_box, has_hand):
        self.keypoints, self.crop_box, self.has_hand = keypoints, crop_box, has_hand
        self.crops_nhwc = torch.zeros((keypoints.shape[0], 176, 176, 4))
        self.range_flags = torch.zeros((4,), dtype=torch.int32)


class _StubNet:
    """what hn_amd.dist.ShardedHandNet drives in the --stub-engine runs"""

    def __init__(self, step):
        self.step = step

    def forward_device(self, images, depth):
        return self.step()


def stub_workload(args, rank):
    """TEST ONLY (--stub-engine): a CPU stand-in for the engine, so that the N-rank plumbing of THIS file runs where there is
    no GPU.  Records are a pure function of (rank, frame); rank r's step sleeps 2 + 0.5 r ms (a visible straggler)."""
    batch = args.batch or 4
    frames = torch.arange(batch, dtype=torch.float32) + 100.0 * rank
    kp = frames.reshape(batch, 1, 1) + torch.arange(63, dtype=torch.float32).reshape(1, 21, 3) * 0.01
    box = torch.stack([frames.to(torch.int64), frames.to(torch.int64) + 1, frames.to(torch.int64) + 30,
                       frames.to(torch.int64) + 40], dim=1)
    has = torch.ones((batch,), dtype=torch.int32)

    def step():
        time.sleep(0.002 + 0.0005 * rank)
        return _StubOutput(kp, box, has)
    info = {"batch_per_gpu": batch, "unit": "frames/s", "gflop_per_unit": 0.0,
            "name": "STUB engine (plumbing test of bench.py on CPU: no GPU work, not a measurement)"}
    return step, info, None


def device_identity(args, local, rank):
    """What this rank computes on: PCI bus id (+ uuid when torch exposes it) of its GPU."""
    if args.stub_engine:
        return f"cpu-stub:{rank}"
    import ctypes
    from hn_amd import _lib
    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.load().hn_device_pci_bus_id(buf, 64), "hn_device_pci_bus_id")     # (the current device = cuda:local)
    p = torch.cuda.get_device_properties(local)
    uuid = getattr(p, "uuid", None)
    return f"{buf.value.decode()} {p.name}" + (f" uuid={uuid}" if uuid is not None else "")


PARITY_CROPS = 16    # crops of the batch-64 A2J step compared with the oracle (other_configs.a2j_b64.parity)
LIVE_PARAS = (617.343, 617.343, 312.42, 241.42)   # SURVEY 8d's intrinsics


def lifter_legs(eng, dev, timed, hip=None):
    """SURVEY 8f #4 and the live caller's chain (ros_demo.py:270-290,329-337): the Pose2Mesh lifter alone at batch 1 and 32
    (hipGraph replay; 23 launches per forward at batch 1), and `live_b1`: HandNet -> clamp + convert_joints in the aggregation's epilogue
    -> lifter input -> Pose2Mesh -> ONE device -> host copy, all of it one captured step on one frame.  The graph hierarchy
    is the data fixture of the lifter's golden test (synthetic topology with the reference's level sizes; the MANO files do
    not travel), weights seeded like every other leg."""
    import numpy as np
    import scipy.sparse as sp
    from hn_amd import synth
    from hn_amd.live import LiveHandEngine
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    f = REPO / "tests" / "golden" / "pose2mesh_forward.npz"
    if not f.exists():
        return {}
    g = np.load(f)
    graphs = [sp.csr_matrix((g[f"L{i}_data"], g[f"L{i}_indices"], g[f"L{i}_indptr"]), shape=tuple(int(v) for v in g[f"L{i}_shape"]))
              for i in range(int(g["num_levels"]))]
    sd = synth.make_pose2mesh_state_dict(0, graph_sizes=[m.shape[0] for m in graphs])
    lifter = Pose2MeshEngine(sd, graphs, device=dev)
    out = {}
    for b, per in ((1, 20), (32, 8)):
        x = torch.randn((b, 21, 2), generator=torch.Generator().manual_seed(5000)).to(dev)
        run, _, _ = lifter.graphed(x)
        rec = timed(run, b, per_group=per, warm=5)
        rec.update(unit="meshes/s", hipgraph=True,
                   launch_structure=("23 launches (6 matrix-vector Linear + 1 glue + 15 fused graph convolutions + fc; "
                                     "profiles/r06b_p2m_kernel_stats.csv)" if b <= Pose2MeshEngine.FUSED_MAX_BATCH else
                                     "layer by layer above 4 samples (throughput-bound)"),
                   workload=f"Pose2Mesh lifter alone, batch {b} (PoseNet MLP + 15 Chebyshev graph convolutions + fc; SURVEY 8f #4)")
        out[f"pose2mesh_b{b}"] = rec
    live = LiveHandEngine(eng, lifter, LIVE_PARAS, clamp=True)
    try:
        rgb1, dep1 = synth.make_rgb(1, seed=1000).to(dev), synth.make_depth(1, seed=2000).to(dev)
        run, _, _, lo = live.graphed(rgb1, dep1)
        rec = timed(run, 1, per_group=10, warm=5)
        torch.cuda.synchronize()
        kp, has, box, _words, (img, xyz), mesh = lo.read()
        if hip is not None:      # what the TIMED captured step copied to the host, for cpu_baseline's oracle chain (live_parity)
            hip["live_b1"] = ((kp, box, img, xyz, mesh), sd, graphs)
        rec.update(unit="frames/s", hipgraph=True,
                   workload="the live caller's chain on one frame (ros_demo.py:270-290,329-337): HandNet -> clamp + convert_joints "
                            "(aggregation epilogue) -> lifter input -> Pose2Mesh -> one device -> host copy, ONE captured step",
                   host_bytes_per_frame=int(lo.host.numel()), mesh_vertices=int(mesh.shape[1]), frames_with_hand=int((has == 1).sum()))
        out["live_b1"] = rec
    finally:
        eng.set_convert(on=False)     # (the run's engine goes back to the plain step: the later legs time that)
    # the same chain without the detector: the stand-alone mesh demo's loop body on one dataset crop (a2j_mesh.py:58-80)
    from hn_amd.live import CropMeshEngine
    cm = CropMeshEngine(eng.a2j, lifter, clamp=True)
    crop1 = synth.make_crops(1, 176, seed=3000).to(dev)
    box1 = torch.tensor([[224.25, 152.5, 400.75, 328.5]], device=dev)
    run, _, _, _, co = cm.graphed(crop1, box1, torch.tensor([LIVE_PARAS], device=dev))
    rec = timed(run, 1, per_group=10, warm=5)
    torch.cuda.synchronize()
    rec.update(unit="crops/s", hipgraph=True, host_bytes_per_crop=int(co.host.numel() * 4),
               workload="the mesh demo's loop body on one dataset crop (a2j_mesh.py:58-80): A2J -> clip + convert_joints with the "
                        "dataset's float32 box and intrinsics (aggregation epilogue) -> lifter input -> Pose2Mesh -> one copy, ONE "
                        "captured step")
    out["crop_mesh_b1"] = rec
    return out


def other_configs_leg(args, info, dev, sds=None, hip=None):
    """BASELINE.json's other single-GPU configurations on the engines this run has already built (10-50 timed steps each, inputs
    resident in HBM, hipGraph replay): config 2 (A2J-only, batch 64; a2j_infer.py:58-60), config 3 (FCOS-only, batch 16;
    trainval_net_fcos.py:124-130,173) and the full pipeline at batch 1 (the reference caller's own batch, ros_demo.py:270,
    as the drop-in runs it).  Each: value, ms_per_step, and the dominant kernel's roofline fraction from
    HIP-event-instrumented eager steps."""
    from hn_amd import synth
    eng = info["engine"]
    terms = 1 if args.precision == "f16x1" else 3
    out = {}
    hip = {} if hip is None else hip    # HIP outputs of the timed steps, for parity_legs()

    def timed(step, units, per_group=4, groups=5, warm=2):
        """`groups` groups of `per_group` steps, each group bracketed by a synchronize; the figure is the MEDIAN group (mean and
        worst group beside it): these legs are a few tens of ms each, and one stall of the fresh box -- seen once: a single
        32 ms pause inside five 3 ms replays -- would otherwise be the number."""
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        per = []
        for _ in range(groups):
            t0 = time.perf_counter()
            for _ in range(per_group):
                step()
            torch.cuda.synchronize()
            per.append(1e3 * (time.perf_counter() - t0) / per_group)
        med = sorted(per)[len(per) // 2]
        return {"value": round(units * 1e3 / med, 2), "ms_per_step": round(med, 3), "steps": per_group * groups,
                "timing": f"median of {groups} groups of {per_group} steps", "ms_per_step_mean": round(sum(per) / len(per), 3),
                "ms_per_step_worst_group": round(max(per), 3)}

    def roof_of(step, ms):
        r = roofline_leg(step, 2, ms, sample_clock=False, terms=terms)
        if r is None:
            return {}
        return {"kernel": r["kernel"], "frac": r["frac"], "achieved_tflops": r["achieved"], "avg_launch_us": r["avg_launch_us"],
                "launches_per_step": r["launches_per_step"], "all_conv_tflops": r["all_conv_achieved"]}

    def captured(fn):
        """hipGraph replay of fn (static launch sequence by construction): the timed figure is the engine's, not the Python
        host's -- the first run on a fresh box has a slow host (image paging in), and five eager steps of ~80 launches each
        measured that instead (16.8 ms per A2J step on one box against 3.1 ms on every other)."""
        from hn_amd import ops
        with torch.inference_mode(False), torch.no_grad(), ops.launch_cost_hidden():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                keep = fn()
        return g, keep

    crops = synth.make_crops(64, 176, seed=3000).to(dev)
    g, keep = captured(lambda: eng.a2j.forward(crops))
    rec = timed(g.replay, 64, per_group=4)
    rec.update(unit="crops/s", hipgraph=True, workload="A2J-only inference, batch 64 176x176 depth crops (BASELINE config 2)",
               gflop_per_unit=round(2 * eng.a2j.macs_per_crop() / 1e9, 3), **roof_of(lambda: eng.a2j.forward(crops), rec["ms_per_step"]))
    out["a2j_b64"] = rec
    # rows 0..15 of the timed batch-64 step's own output: compared with the oracle by cpu_baseline() AFTER every timed region
    hip["a2j_b64"] = (keep[:PARITY_CROPS].cpu(), crops[:PARITY_CROPS].cpu())
    del crops, g, keep
    rgb16 = synth.make_rgb(16, seed=1000).to(dev)
    g, keep = captured(lambda: eng.fcos.detect(rgb16))
    rec = timed(g.replay, 16, per_group=2)
    rec.update(unit="frames/s", hipgraph=True, workload="FCOS ResNet34-FPN detector, batch 16 640x480 RGB (BASELINE config 3)",
               gflop_per_unit=round(2 * eng.fcos.macs_per_frame() / 1e9, 3), **roof_of(lambda: eng.fcos.detect(rgb16), rec["ms_per_step"]))
    out["fcos_b16"] = rec
    det = keep[0]     # the timed batch-16 step's own detections (score-ordered survivors + their candidate indices)
    cnt = det.count.cpu().tolist()
    bx, sc, lb, kp_ = det.boxes.cpu(), det.scores.cpu(), det.labels.cpu(), det.keep.cpu()
    hip["fcos_b16"] = [(bx[i, :k].clone(), sc[i, :k].clone(), lb[i, :k].clone(), kp_[i, :k].clone()) for i, k in enumerate(cnt)]
    del rgb16, g, keep, det
    rgb1, dep1 = synth.make_rgb(1, seed=1000).to(dev), synth.make_depth(1, seed=2000).to(dev)
    run, _, _, _ = eng.graphed(rgb1, dep1)
    rec = timed(run, 1, per_group=10, warm=5)
    rec.update(unit="frames/s", hipgraph=True,
               workload="Full HandNet pipeline at batch 1 (the reference caller's batch, ros_demo.py:270), hipGraph replay",
               **roof_of(lambda: eng.forward_device(rgb1, dep1), rec["ms_per_step"]))
    out["pipeline_b1"] = rec
    if args.precision == "f16x3":
        out.update(lifter_legs(eng, dev, timed, hip))
    if args.precision == "f16x3" and sds is not None:
        # the reference's own arithmetic -- IEEE fp32 operands on the f32 MFMA -- driver-timed beside the split-fp16 headline:
        # the same step on engines built with precision="f32", eager, 3 timed steps; roofline against the f32-MFMA peak
        from hn_amd.a2j_engine import A2JEngine
        from hn_amd.fcos_engine import FCOSEngine
        from hn_amd.pipeline import HandNetEngine
        fcos_sd, a2j_sd = sds
        e32 = HandNetEngine(FCOSEngine(fcos_sd, 3, device=dev, precision="f32"), A2JEngine(a2j_sd, device=dev, precision="f32"), 3)
        rgb, depth = synth.make_rgb(32, seed=1000).to(dev), synth.make_depth(32, seed=2000).to(dev)
        rec = timed(lambda: e32.forward_device(rgb, depth), 32, per_group=1, groups=3, warm=1)
        rec.update(unit="frames/s", hipgraph=False, dtype="f32 (IEEE fp32 operands, f32 MFMA, fp32 accumulate)",
                   workload="Full HandNet pipeline, batch 32, exact-f32 engines (precision='f32'): BASELINE config 4 in the "
                            "reference's own arithmetic", peak_tflops=F32_MFMA_PEAK_TFLOPS,
                   **roof_of(lambda: e32.forward_device(rgb, depth), rec["ms_per_step"]))
        out["pipeline_b32_f32"] = rec
        del e32, rgb, depth
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "HN_BENCH_WORKER" not in os.environ:
        sys.exit(supervise(args))
    worker(args)


def worker(args):
    from hn_amd import dist as hdist
    multi = args.gpus > 1
    attempt = int(os.environ.get("HN_BENCH_ATTEMPT", "0"))
    job_dir = Path(os.environ["HN_BENCH_DIR"]) if multi else None
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report a {world_env}-rank run as "
                         f"{args.gpus} GPUs")
    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank_env = int(os.environ.get("RANK", "0"))
    devices = None
    try:
        if multi and os.environ.get("HN_BENCH_INJECT_INIT_FAILURE") in (f"{rank_env}:{attempt}", f"{rank_env}:*"):
            # (tests: what a host whose IPC-handle mode does not match looks like to this file)
            raise RuntimeError("injected for the ladder test: hipIpcGetMemHandle: invalid argument")
        rank, local, world = hdist.init_from_env(
            args.dist_backend if multi else None,
            init_method=f"file://{job_dir}/store{attempt}" if multi else None, timeout_s=args.init_timeout if multi else None)
        if not args.stub_engine:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
            torch.cuda.set_device(local)
        if multi:
            # the first collective: every rank's device, on every rank (also what proves that the group carries data)
            devices = [None] * world
            dist.all_gather_object(devices, device_identity(args, local, rank))
            (job_dir / f"attempt{attempt}.rank{rank}.ready").write_text("up\n")
    except SystemExit:
        raise
    except Exception as e:  # noqa: BLE001 -- RCCL / rendezvous failure: report it; the supervisor decides about the second rung
        msg = (f"rank {rank_env}: torch.distributed ({args.dist_backend}) failed to initialise for --gpus {args.gpus} with "
               f"{os.environ.get('HN_BENCH_IPC_MODE', _ipc_label(os.environ))}: {type(e).__name__}: {e}")
        print(f"[bench] {msg}", file=sys.stderr, flush=True)
        if job_dir is not None:
            try:
                with open(job_dir / f"attempt{attempt}.failed", "x") as f:
                    f.write(msg[:2000] + "\n")
            except OSError:
                pass
        sys.exit(EXIT_INIT_FAILED)
    if multi and os.environ.get("HN_BENCH_INJECT_RUN_FAILURE") == str(rank_env):
        raise SystemExit("[bench] injected for the supervisor test: a failure after the process group was up")
    if multi and not args.share_gpu and not args.stub_engine and len(set(devices)) != world:
        raise SystemExit(f"[bench] {world} ranks but only {len(set(devices))} distinct devices: {devices}")
    dev = torch.device("cpu") if args.stub_engine else torch.device("cuda", local)
    # development host: the HN_* A/B variables of tools/ select older kernel forms / launch structures for same-box comparisons
    # (hn_amd/forms.py).  The product itself never reads them; a run that used any reports them in config.forms.
    forms_used = None
    if not args.stub_engine:
        from hn_amd import forms
        forms_used = forms.apply_env()
    if world > 1:   # N ranks build their engines at once on one host: share the cores instead of oversubscribing them N-fold
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
        if args.workload == "pipeline" and not args.eager and not args.native:
            args.graph = True   # N Python hosts on one box: replay the step instead of issuing ~150-200 launches per rank from Python

    if args.stub_engine:
        step, info, sds = stub_workload(args, rank)
    else:
        step, info, sds = build_workload(args, dev, rank)
    batch = info["batch_per_gpu"]
    on_gpu = not args.stub_engine

    gathered = {"rows": None}
    ev = {"pairs": [], "on": False}
    sharded = None
    if world > 1 and args.workload == "pipeline":
        # the product's N > 1 callable (hn_amd.dist.ShardedHandNet): this rank's resident frames as its shard of the global
        # batch; the step AND the all-gather of the per-frame records replay from ONE hipGraph when RCCL lets itself be
        # captured (config.gather says which)
        if args.stub_engine:
            sharded = hdist.ShardedHandNet(_StubNet(step), use_graph=False)
            shard_in = (torch.zeros((batch, 3, 2, 2)), torch.zeros((batch, 1, 2, 2)))
        elif not args.native:
            sharded = hdist.ShardedHandNet(info["engine"], use_graph=(True if args.capture_gather else "step") if args.graph else False)
            shard_in = info["inputs"]
            if args.graph:
                sharded.prepare(*shard_in, global_batch=world * batch)

    def full_step():
        if ev["on"] and on_gpu:     # HIP events (torch's current stream = the launch stream) around THIS rank's engine work
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        t_a = time.perf_counter()
        if sharded is not None:
            out = sharded.forward_device(*shard_in, global_batch=world * batch)
            gathered["rows"] = out.valid
        else:
            out = step()
        if ev["on"]:
            if on_gpu:
                b.record()
                ev["pairs"].append((a, b))
            else:
                ev["pairs"].append(time.perf_counter() - t_a)
        if sharded is None and world > 1 and args.workload == "pipeline":
            if args.dist_backend == "gloo" and on_gpu:  # rehearsal only: gloo gathers host tensors
                g = hdist.gather_results(out.keypoints.cpu(), out.crop_box.cpu(), out.has_hand.cpu(), per_rank=batch)
            else:
                g = hdist.gather_results(out.keypoints, out.crop_box, out.has_hand, per_rank=batch)
            gathered["rows"] = g[3]
        return out

    def fence():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        full_step()
    fence()
    ev["on"] = world > 1
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    fence()
    elapsed = time.perf_counter() - t0
    ev["on"] = False
    rank_ms = None
    if world > 1:
        on_dev = args.dist_backend == "nccl" and on_gpu
        t = torch.tensor([elapsed], device=dev if on_dev else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own engine time per step (HIP events; without the gather and without waiting for the others)
        if sharded is not None:
            # the product callable's step holds the collective (inside its hipGraph when RCCL allows), i.e. every rank's step
            # waits for the slowest: a straggler is only visible in the engine-only step, timed here AFTER the timed region
            k = max(1, min(args.steps, 5))
            if on_gpu:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                step()
                torch.cuda.synchronize()
                a.record()
                for _ in range(k):
                    step()
                b.record()
                torch.cuda.synchronize()
                mine = a.elapsed_time(b) / k
            else:
                t_a = time.perf_counter()
                for _ in range(k):
                    step()
                mine = 1e3 * (time.perf_counter() - t_a) / k
        else:
            mine = (sum(a.elapsed_time(b) for a, b in ev["pairs"]) if on_gpu else 1e3 * sum(ev["pairs"])) / max(1, args.steps)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, float(mine))
        srt = sorted(per_rank)
        rank_ms = {"min_median_max": [round(srt[0], 3), round(srt[len(srt) // 2] if len(srt) % 2 else
                                      0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2]), 3), round(srt[-1], 3)],
                   "per_rank": [round(v, 3) for v in per_rank],
                   "what": "engine launches of one step on each rank's own stream (HIP events), without the all-gather"
                           + (" (timed after the timed region: the product callable's step holds the collective)" if sharded is not None else "")}

    # what the collective really spanned: ranks seen by an actual all_gather_into_tensor (not the --gpus flag)
    rccl_ranks = world
    if world > 1 and gathered["rows"] is not None:
        rccl_ranks = int(gathered["rows"].numel()) // batch
        if rccl_ranks != world or not bool(gathered["rows"].all()):
            raise SystemExit(f"all-gather returned {gathered['rows'].numel()} rows for {world} ranks x {batch} frames")
    roof = None
    if on_gpu and not args.no_roofline and not args.native and (not args.graph or "eager_step" in info):
        roof = roofline_leg(info.get("eager_step", step) if args.graph else step, max(1, min(args.steps, 3)),
                            1e3 * elapsed / args.steps, not args.no_clock_sample, terms=1 if args.precision == "f16x1" else 3)
    single = rank == 0 and world == 1 and on_gpu
    dropin = None
    if single and args.workload == "pipeline" and not args.no_dropin and not args.native and not args.graph:
        dropin = dropin_leg(args, sds, dev, batch)
    others, hip_outputs = None, None
    if single and args.workload == "pipeline" and not args.no_other_configs and not args.native and not args.graph:
        hip_outputs = {}
        others = other_configs_leg(args, info, dev, sds, hip_outputs)
    cpu = None
    if single and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sds, info.get("engine"), dev, others, hip_outputs)

    if rank == 0:
        units = world * batch * args.steps
        value = units / elapsed
        why_file = job_dir / f"attempt{attempt}.why" if multi else None
        ipc_mode = os.environ.get("HN_BENCH_IPC_MODE", _ipc_label(os.environ))
        if multi:
            ipc_mode += f" (attempt {attempt + 1} of 2" + (f"; the first set failed to initialise: {why_file.read_text().strip()}"
                                                         if why_file.exists() else "") + ")"
        line = {
            "metric": "end-to-end frames/sec (FCOS+A2J, 640x480)" if args.workload == "pipeline"
            else f"{args.workload} throughput",
            "value": round(value, 2), "unit": info["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "f16x3": "f16x3 (fp32 values split into fp16 hi+lo, 3 MFMAs, fp32 accumulate; fp32 I/O)",
                      "f16x1": "f16x1 (THROUGHPUT MODE, not parity-grade: fp16 hi parts only, 1 MFMA per MAC, fp32 accumulate; "
                               "error figures in cpu_baseline.parity)"}[args.precision],
            "data": "synthetic (seeded uniform RGB in [0,1), depth 0.3-1.5 m; random-init weights of the "
                    "reference architectures, hn_amd.synth seed 0)",
            "config": {"workload": info["name"], "batch_per_gpu": batch, "global_batch": batch * world,
                       "frame": {"a2j": "176x176 depth crop", "pose2mesh": "21 x 2-D joints"}.get(args.workload, "640x480 RGB-D"),
                       "parallelism": f"frames sharded over {world} GPU(s), one all-gather of per-frame records per step",
                       "collective_backend": (dist.get_backend() if world > 1 else None), "rccl_ranks": rccl_ranks,
                       "gather": (sharded.capture_note if sharded is not None and args.graph and on_gpu else
                                  "eager all-gather behind the step" if world > 1 else None),
                       "devices": devices, "ipc_mode": ipc_mode,
                       "gflop_per_unit": round(info["gflop_per_unit"], 3), "hipgraph": bool(args.graph),
                       "host": "STUB" if args.stub_engine else "C++ layer graph (model-level C ABI)" if args.native
                       else "Python engines (op-level C ABI)",
                       "forms": forms_used or None},
            "algorithmic_tflops": round(value * info["gflop_per_unit"] / 1e3, 2),
            "rank_ms": rank_ms,
            "roofline": roof, "other_configs": others, "dropin": dropin, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Throughput benchmark of the FCOS -> crop -> A2J hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms run N ranks, one per GPU: started without a torchrun environment and with --gpus > 1, this
file launches `python -m torch.distributed.run` on itself BEFORE anything touches the GPU (the parent
never initialises HIP, relays the children's output and exits with their status).  A world size that
differs from --gpus is an error, never a silent single-GPU run.

One "step" = one pass of HandNet (FCOS detector, top-1 hand crop, A2J) over one batch of
synthetic 640x480 RGB-D frames already resident in HBM, plus -- for N > 1 -- the all-gather
of the per-frame results (RCCL).  Weak scaling: every rank processes its own `--batch`
frames (BASELINE.json config 4: 32 frames on one GPU; config 5: 8 x 32).  Rank 0 prints ONE
JSON line; `value` is whole-job frames/s = N * batch * K / max-over-ranks(time).

Extra objects in the line:
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
                CPU) -- at the bench batch and at batch 1 (the only batch size the reference's caller uses).
  cpu_baseline  the oracle (CPU restatement, kind "port") timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N == 1 only); the same frames go
                through the HIP engine and the agreement is reported as `parity`.
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
    ap.add_argument("--workload", choices=["pipeline", "a2j", "fcos", "pose2mesh"], default="pipeline")
    ap.add_argument("--precision", choices=["f16x3", "f32", "f16x1"], default="f16x3",
                    help="f16x3: split-fp16 operands on the f16 MFMA (fp32-grade results; the headline); f32: exact f32 MFMA; "
                         "f16x1: hi*hi term only = plain fp16 operands, 1 MFMA per MAC -- the THROUGHPUT mode SURVEY D6 plans "
                         "beside the parity mode: misses the 1e-3 contract, reported with its error figures, never the headline")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
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
    batch = args.batch or (64 if wl == "a2j" else 32 if wl in ("pipeline", "pose2mesh") else 16)
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
                      "csrc/conv_igemm_f16x3.hip (data movement may differ)")
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
        # template arguments <BM, BN, WM, WN, NBUF, BUF, RS, POOL, TERMS> (f16x3) / <BM, BN, SMALLC> (f32)
        targs = [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")]
        if prec == "f16x3" and (len(targs) < 9 or (targs[6] == "true") != rs or targs[7] != "false" or targs[8] != "3"):
            continue
        return k["hbm_bytes_per_launch"], f"profiles/{rec.get('tag')}_traffic.json (commit {rec.get('commit')})"
    return None, f"{rec.get('tag')}: no {needle} record"


def kernel_source_sha16():
    """Identity of the dominant kernel's source: the HBM-traffic figure of an older revision is never paired with a run."""
    import hashlib
    return hashlib.sha256((REPO / "handnet-pipeline_amd" / "csrc" / "conv_igemm_f16x3.hip").read_bytes()).hexdigest()[:16]


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
    del net
    return out


def cpu_baseline(args, sds, engine=None, dev=None):
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
    stats, oracle_s, (g_kp, g_box, g_has, o_kp, o_box, o_has) = parity.pipeline_parity(
        engine, rgb, depth, fcos_sd, a2j_sd, 3, chunk=8)
    res = {"value": round(n / oracle_s, 3), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{n} frames (seeds 1000/2000) in batches of 8, full-pipeline oracle forward "
                     "(torch CPU fp32, FCOS+crop+A2J)"}
    eq = (g_box == o_box).all(dim=1) & g_has & o_has
    if bool(eq.any()):
        paras = (617.343, 617.343, 312.42, 241.42)   # SURVEY 8d intrinsics for the millimetre figure
        valid = eq.to(torch.int32).to(dev)
        xyz = ops.convert_joints(g_kp.to(dev), g_box.to(dev), valid, paras).cpu()
        xyz_ref = ops.convert_joints(o_kp.to(dev), o_box.to(dev), valid, paras).cpu()
        stats["mm_epe"] = float((xyz - xyz_ref)[eq].norm(dim=-1).mean())
    res["parity"] = stats
    return res


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start one rank per GPU with torch.distributed.run and
    relay its output.  Runs BEFORE anything initialises HIP in this process (no torch.cuda.is_available(), no
    library load): the parent only waits for the children and exits with their status -- it is never replaced by
    another program."""
    have = torch.cuda.device_count()          # does not initialise the GPU on this image
    if not args.share_gpu and have < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) are visible "
                         "(--share-gpu --dist-backend gloo rehearses the plumbing on one)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd).returncode)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    from hn_amd import dist as hdist
    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    try:
        rank, local, world = hdist.init_from_env(args.dist_backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    except Exception as e:  # noqa: BLE001 -- RCCL / rendezvous failure: report it and stop; never a fallback, never a re-exec
        raise SystemExit(f"[bench] rank {os.environ.get('RANK', '0')}: torch.distributed ({args.dist_backend}) failed to "
                         f"initialise for --gpus {args.gpus}: {type(e).__name__}: {e}")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as "
                         f"{args.gpus} GPUs")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # development host: the HN_* A/B variables of tools/ select older kernel forms / launch structures for same-box comparisons
    # (hn_amd/forms.py).  The product itself never reads them; a run that used any reports them in config.forms.
    from hn_amd import forms
    forms_used = forms.apply_env()
    if world > 1:   # N ranks build their engines at once on one host: share the cores instead of oversubscribing them N-fold
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))

    step, info, sds = build_workload(args, dev, rank)
    batch = info["batch_per_gpu"]

    gathered = {"rows": None}

    def full_step():
        out = step()
        if world > 1 and args.workload == "pipeline":
            if args.dist_backend == "gloo":  # rehearsal only: gloo gathers host tensors
                g = hdist.gather_results(out.keypoints.cpu(), out.crop_box.cpu(), out.has_hand.cpu(), per_rank=batch)
            else:
                g = hdist.gather_results(out.keypoints, out.crop_box, out.has_hand, per_rank=batch)
            gathered["rows"] = g[3]
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        full_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what the collective really spanned: ranks seen by an actual all_gather_into_tensor (not the --gpus flag)
    rccl_ranks = world
    if world > 1 and gathered["rows"] is not None:
        rccl_ranks = int(gathered["rows"].numel()) // batch
        if rccl_ranks != world or not bool(gathered["rows"].all()):
            raise SystemExit(f"all-gather returned {gathered['rows'].numel()} rows for {world} ranks x {batch} frames")
    roof = None
    if not args.no_roofline and not args.graph and not args.native:
        roof = roofline_leg(step, max(1, min(args.steps, 3)), 1e3 * elapsed / args.steps, not args.no_clock_sample,
                            terms=1 if args.precision == "f16x1" else 3)
    dropin = None
    if (rank == 0 and world == 1 and args.workload == "pipeline" and not args.no_dropin and not args.native
            and not args.graph):
        dropin = dropin_leg(args, sds, dev, batch)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sds, info.get("engine"), dev)

    if rank == 0:
        units = world * batch * args.steps
        value = units / elapsed
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
                       "gflop_per_unit": round(info["gflop_per_unit"], 3), "hipgraph": bool(args.graph),
                       "host": "C++ layer graph (model-level C ABI)" if args.native else "Python engines (op-level C ABI)",
                       "forms": forms_used or None},
            "algorithmic_tflops": round(value * info["gflop_per_unit"] / 1e3, 2),
            "roofline": roof, "dropin": dropin, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Parity sweep: the drop-in HandNet on HIP against the oracle away from the single (seed-0 weights, white-noise frames)
operating point of the other end-to-end tests.

Cases (tests/parity_cases.py): other weight draws; structured frames (black / constant / saturated / ramps / checkerboard /
low-pass / one rectangle) mixed INTO one batch with noise frames; depth maps with zero holes and a constant depth map;
detectors whose output biases are raised until one image yields > 1000 and > 4000 candidates (torchvision's per-class
batched_nms branch, fcos_utils/fcos.py:635) and lowered until a batch mixes frames with and without a hand and frames with
no candidate at all.  Every case also runs with the range check on (HN_CHECK_RANGE=1's flag).

What is asserted per frame, against the oracle on the same inputs (delta = max |HIP score - oracle score| over ALL 17 850
anchor points of the frame, itself asserted <= 3e-5; measured 1.3e-5 -- the fp32 oracle is 6e-5..1e-4 from its own fp64 run):
  candidate set (anchor-point indices that pass `scores_max > 0.7`, fcos.py:600) ......... identical
  NMS survivors: the REFERENCE batched_nms, fed the oracle's boxes and labels in the HIP engine's score order, returns
      exactly the HIP survivor list (indices and order) .................................. always (bit-exact NMS)
  survivor list vs the oracle's own ...................................................... identical, or differing only
      through pairs of candidates whose oracle scores are closer than 2 delta (near / exact score TIES: the order of two
      scores that close is not determined by fp32-grade arithmetic -- the reference's own CPU and CUDA runs would not agree
      on it either); such frames are counted and reported ("order" frames)
  top-1 hand crop box (int64 truncation + 0.4 padding + clamp, handnet_pipeline.py:88-97) identical (else diagnosed)
  depth crops (pure gather) ............................................................. identical
  keypoints, EVERY frame with a hand ..................................................... |d| < max(1e-3, 3 x the
      oracle's own fp32-vs-fp64 difference on the same crops), both printed; where a tolerated decision moved the crop box the
      oracle's A2J runs on the HIP engine's crop, so no frame escapes the keypoint comparison
A frame whose candidate set or crop box differs is never waved through: `_diagnose` re-runs the oracle for that frame in
fp64 and the frame is tolerated only if the deciding quantity (a score against 0.7, a box coordinate against an integer)
sits inside max(2 delta, 3 x the fp32 oracle's distance from its fp64 self).  The offending values are printed and
collected (gpurun_out/parity_sweep_report.json, copied to profiles/); `test_tolerated_frames_are_rare` bounds their number.
"""
import json
import os
import types
from pathlib import Path

import pytest
import torch

import parity_cases as pc

pytestmark = pytest.mark.gpu

NUM_CLASSES = 3
HAND = NUM_CLASSES - 1
KP_TOL = 1e-3                     # north_star: keypoints within 1e-3 of the fp32 reference
REPORT = {"cases": {}, "tolerated": [], "order": []}
# these cases also run on the EXACT f32-MFMA engines (precision="f32"), to see whether the near-tied score
# pairs that the split-fp16 path ranks differently from the oracle are flipped by fp32 summation order alone
F32_CASES = ["seed1", "structured", "cands_1100"]
_FLIPS = {}      # (case, mode) -> {frame: set of (candidate a, candidate b) pairs ranked differently from the oracle}
_ENG32 = {}


# ---------------------------------------------------------------------------------------------------------------------
# cases
# ---------------------------------------------------------------------------------------------------------------------
def _interleave(a, b):
    out = []
    for i in range(max(len(a), len(b))):
        out += ([a[i]] if i < len(a) else []) + ([b[i]] if i < len(b) else [])
    return out


def _case(name):
    """-> dict(weights=(fcos key, a2j seed), frames=list of [3,H,W], depth=[N,1,H,W], names=list of str)"""
    if name.startswith("seed"):
        s = int(name[4:])
        rgb = pc.noise_frames(16, 1000 + s)
        return dict(weights=(("seed", s), s), frames=list(rgb), depth=pc.depth_noise(16, 2000 + s))
    if name == "structured":
        st = pc.structured_frames()
        noise = pc.noise_frames(8, 1100)
        frames = _interleave(list(st.values()), list(noise))
        names = _interleave(list(st.keys()), [f"noise{i}" for i in range(8)])
        return dict(weights=(("seed", 0), 0), frames=frames, depth=pc.depth_noise(len(frames), 2100), names=names)
    if name == "depth_holes":
        return dict(weights=(("seed", 0), 0), frames=list(pc.noise_frames(8, 1200)), depth=pc.depth_with_holes(8, 2200))
    if name == "depth_constant":
        return dict(weights=(("seed", 0), 0), frames=list(pc.noise_frames(8, 1200)), depth=pc.depth_constant(8, 0.75))
    if name == "cands_1100":     # K > 1000: boxes.numel() > 4000, the per-class branch of batched_nms
        return dict(weights=(("shift", 1.0), 0), frames=list(pc.noise_frames(4, 1300)), depth=pc.depth_noise(4, 2300))
    if name == "cands_5200":     # past the 2048-key LDS sort and the 4096-key path of the NMS kernel
        return dict(weights=(("shift", 2.2), 0), frames=list(pc.noise_frames(2, 1400)), depth=pc.depth_noise(2, 2400))
    if name == "mixed_hands":    # ~15 candidates per frame: frames with and without a hand in ONE batch (10 of 16 with)
        return dict(weights=(("shift", -1.5), 0), frames=list(pc.noise_frames(16, 1500)), depth=pc.depth_noise(16, 2500))
    if name == "sparse_hands":   # 4 of 16 frames with a hand: the engine's sparse-stream path (A2J on those frames only)
        return dict(weights=(("shift", -1.75), 0), frames=list(pc.noise_frames(16, 1500)), depth=pc.depth_noise(16, 2500))
    if name == "frames_360x480":   # another source size: other interpolation weights, other clamps of the crop rule (W = 480, H = 360)
        return dict(weights=(("seed", 1), 1), frames=list(pc.noise_frames(8, 1700, 360, 480)),
                    depth=pc.depth_noise(8, 2700, 360, 480))
    if name == "frames_720x1280":  # upscale by 1.04 (max side 1333 binds): the <16, 160> LDS patch of the tiled preprocess, a 768 x 1344 canvas
        return dict(weights=(("seed", 1), 1), frames=list(pc.noise_frames(4, 1900, 720, 1280)),
                    depth=pc.depth_noise(4, 2900, 720, 1280))
    if name == "frames_1080x1920":  # DOWNscale by 0.69: beyond the LDS patches, the per-pixel preprocess kernel
        return dict(weights=(("seed", 1), 1), frames=list(pc.noise_frames(2, 1950, 1080, 1920)),
                    depth=pc.depth_noise(2, 2950, 1080, 1920))
    if name == "rgbd":             # the RGB-D A2J variant (handnet_pipeline.py:99-102: 4-channel crops, channel order [2,1,0,3])
        fr = pc.noise_frames(8, 1850)
        return dict(weights=(("seed", 2), ("rgbd", 3)), frames=list(fr), depth=torch.cat([fr, pc.depth_noise(8, 2850)], dim=1), rgbd=True)
    if name == "two_classes":      # the constructor default num_classes = 2 (handnet_pipeline.py:47): hand class 1, Cout = 4 outputs
        return dict(weights=(("seed2c", 2), 2), frames=list(pc.noise_frames(8, 1800)), depth=pc.depth_noise(8, 2800), classes=2)
    if name == "no_candidates":  # most frames without a single candidate
        return dict(weights=(("shift", -2.5), 0), frames=list(pc.noise_frames(8, 1600)), depth=pc.depth_noise(8, 2600))
    raise KeyError(name)


CASES = ["seed1", "seed2", "seed3", "structured", "depth_holes", "depth_constant", "cands_1100", "cands_5200",
         "mixed_hands", "sparse_hands", "no_candidates", "frames_360x480", "frames_720x1280", "frames_1080x1920", "two_classes", "rgbd"]

_FCOS_SD, _A2J_SD, _NETS, _ORACLE, _FACTS = {}, {}, {}, {}, {}


def _fcos_sd(key):
    if key not in _FCOS_SD:
        from hn_amd import synth
        kind, v = key
        if kind == "seed2c":
            _FCOS_SD[key] = synth.make_fcos_state_dict(v, 2)
            return _FCOS_SD[key]
        base = synth.make_fcos_state_dict(v if kind == "seed" else 0, NUM_CLASSES)
        _FCOS_SD[key] = base if kind == "seed" else pc.shift_detector_bias(base, cls_shift=v, num_classes=NUM_CLASSES)
    return _FCOS_SD[key]


def _a2j_sd(seed):
    """seed, or ("rgbd", seed) for the 4-channel stem of the RGB-D variant"""
    if seed not in _A2J_SD:
        from hn_amd import synth
        _A2J_SD[seed] = synth.make_a2j_state_dict(seed[1], rgbd=True) if isinstance(seed, tuple) else synth.make_a2j_state_dict(seed)
    return _A2J_SD[seed]


def _rgbd(weights):
    return isinstance(weights[1], tuple)


def _classes(weights):
    return 2 if weights[0][0] == "seed2c" else NUM_CLASSES


def _net(weights):
    if weights not in _NETS:
        from handnet_pipeline.handnet_pipeline import HandNet
        args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j="unused.pth")
        if _rgbd(weights):   # the reference builds its RGB-D model from a Lightning checkpoint (handnet_pipeline.py:28-29)
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                args.pretrained_a2j = os.path.join(tmp, "a2j_rgbd.ckpt")
                torch.save({"state_dict": {"a2j." + k: v for k, v in _a2j_sd(weights[1]).items()},
                            "hyper_parameters": {"num_classes": 21, "is_RGBD": True}}, args.pretrained_a2j)
                net = HandNet(args, reload_detector=False, num_classes=_classes(weights), reload_a2j=True, RGBD=True)
        else:
            net = HandNet(args, reload_detector=False, num_classes=_classes(weights), reload_a2j=False, RGBD=False)
            net.a2j.load_state_dict(_a2j_sd(weights[1]), strict=False)
        net.detector.load_state_dict(_fcos_sd(weights[0]), strict=False)
        _NETS.clear()                  # one resident engine pair at a time
        _NETS[weights] = net.cuda().eval()
    return _NETS[weights]


def _oracle(name):
    """Oracle results of a case (computed once; the range-check pass reuses them)."""
    if name in _ORACLE:
        return _ORACLE[name]
    from oracle import a2j_ref, fcos_ref, handnet_ref
    c = _case(name)
    fsd, asd = _fcos_sd(c["weights"][0]), _a2j_sd(c["weights"][1])
    frames, depth = c["frames"], c["depth"]
    dets, cands, smax = [], [], []
    for lo in range(0, len(frames), 4):
        d, inter = fcos_ref.fcos_forward(frames[lo:lo + 4], fsd, _classes(c["weights"]), return_intermediates=True)
        dets += d
        cands += inter["candidates"]
        ho = inter["head"]
        smax += list(torch.sqrt(torch.sigmoid(ho["cls_logits"]) * torch.sigmoid(ho["bbox_ctrness"])).max(dim=-1)[0])
    rgbd = _rgbd(c["weights"])
    cin = 4 if rgbd else 1
    mask, boxes, dcrops = handnet_ref.select_and_crop(dets, depth, _classes(c["weights"]), rgbd)
    kp = torch.zeros((len(frames), 21, 3))
    noise64 = 0.0
    if dcrops:
        x = torch.stack(dcrops)
        k32 = a2j_ref.a2j_forward(x, asd, channel_in=cin)
        k64 = a2j_ref.a2j_forward(x, a2j_ref.to_dtype(asd, torch.float64), dtype=torch.float64, channel_in=cin)
        noise64 = float((k32.double() - k64).abs().max())
        kp[mask] = k32
    # the reference's return tuple (handnet_pipeline.py:107-116), assembled exactly as oracle/handnet_ref.handnet_forward does
    ref_tuple = (kp, torch.stack(dcrops), torch.stack(boxes)) if dcrops else \
        (kp, torch.zeros_like(depth), torch.zeros((len(frames), 4)))
    _FACTS[name] = dict(candidates=[len(x["index"]) for x in cands], mask=mask.tolist(), names=c.get("names"))
    _ORACLE.clear()
    _ORACLE[name] = dict(case=c, dets=dets, cands=cands, mask=mask, boxes=boxes, dcrops=dcrops, kp=kp, noise64=noise64,
                         ref_tuple=ref_tuple, smax=smax)
    return _ORACLE[name]


# ---------------------------------------------------------------------------------------------------------------------
# the HIP side's scores of ALL anchor points, and the fp64 diagnosis of a frame whose candidate set or crop box differs
# ---------------------------------------------------------------------------------------------------------------------
SCORE_TOL = 3e-5      # max |HIP score - oracle fp32 score| over all points of a frame (measured: 1.3e-5)


def _hip_scores(eng, batch):
    """scores_max [N, P] of every anchor point from the HIP engine's own head tensors (fcos.py:593-598 on the device)"""
    cls_lr, reg_ctr, _, _ = eng.fcos.forward_heads(batch)
    n = batch.shape[0]
    cls = torch.cat([t.reshape(n, -1, t.shape[-1])[..., :eng.num_classes] for t in cls_lr], dim=1)
    ctr = torch.cat([t.reshape(n, -1, t.shape[-1])[..., 4:5] for t in reg_ctr], dim=1)
    return torch.sqrt(torch.sigmoid(cls) * torch.sigmoid(ctr)).max(dim=-1)[0].cpu()


def _oracle64(ora, i):
    """the oracle for frame i in fp64: (detections, all-point scores)"""
    from oracle import a2j_ref, fcos_ref
    c = ora["case"]
    fsd = _fcos_sd(c["weights"][0])
    d64, i64 = fcos_ref.fcos_forward([c["frames"][i].double()], a2j_ref.to_dtype(fsd, torch.float64), _classes(c["weights"]),
                                     return_intermediates=True)
    ho = i64["head"]
    return d64[0], torch.sqrt(torch.sigmoid(ho["cls_logits"][0]) * torch.sigmoid(ho["bbox_ctrness"][0])).max(dim=-1)[0]


def _diagnose_candidates(name, i, ora, hip_points, s_hip, delta):
    """candidate sets differ: every point that changed sides must sit on the 0.7 threshold (fcos.py:600)"""
    s32 = ora["smax"][i].double()
    _, s64 = _oracle64(ora, i)
    noise = float((s32 - s64).abs().max())
    sym = sorted(set(hip_points.tolist()) ^ set(ora["cands"][i]["index"].tolist()))
    margin = max(2.0 * delta, 3.0 * noise)
    worst = max(abs(float(s32[p]) - 0.7) for p in sym)
    rec = {"case": name, "frame": i, "kind": "candidate set (score vs 0.7, fcos.py:600)", "hip_vs_oracle_score_delta": delta,
           "oracle_fp32_vs_fp64_score_noise": noise, "decision_margin": margin, "worst_margin_to_0.7": worst,
           "points": [{"point": q, "oracle_fp32": float(s32[q]), "oracle_fp64": float(s64[q]), "hip": float(s_hip[q])}
                      for q in sym]}
    return worst <= margin, rec


def _diagnose_crop(name, i, ora, hip_box, delta):
    """same survivors, different integer crop box: a coordinate of the top-1 hand box straddles an integer
    (handnet_pipeline.py:88-97)"""
    from oracle import handnet_ref
    d = ora["dets"][i]
    d64, _ = _oracle64(ora, i)
    hand = _classes(ora["case"]["weights"]) - 1
    hb = d["boxes"][d["labels"] == hand][:1]
    hb64 = d64["boxes"][d64["labels"] == hand][:1]
    rec = {"case": name, "frame": i, "kind": "crop box (int64 truncation, handnet_pipeline.py:88)",
           "oracle_fp32_top_hand_box": hb.tolist(), "oracle_fp64_top_hand_box": hb64.tolist(), "hip_crop": hip_box.tolist(),
           "oracle_crop": handnet_ref.crop_box(hb, ora["case"]["depth"].shape[-1], ora["case"]["depth"].shape[-2]).tolist()
           if len(hb) else None}
    if not len(hb) or not len(hb64):
        return False, rec
    coord_noise = float((hb.double() - hb64).abs().max())
    # distance of a coordinate, and of its 0.4-padded self, to the next integer
    x = hb.reshape(4).double()
    w, h = torch.trunc(x[2]) - torch.trunc(x[0]), torch.trunc(x[3]) - torch.trunc(x[1])
    padded = torch.stack([torch.trunc(x[0]) - 0.4 * w, torch.trunc(x[1]) - 0.4 * h, torch.trunc(x[2]) + 0.4 * w,
                          torch.trunc(x[3]) + 0.4 * h])
    dist = min(float((x - x.round()).abs().min()), float((padded - padded.round()).abs().min()))
    rec.update(oracle_fp32_vs_fp64_coordinate_noise=coord_noise, min_distance_to_integer=dist)
    return dist <= max(3.0 * coord_noise, 1e-3), rec


# ---------------------------------------------------------------------------------------------------------------------
# the sweep
# ---------------------------------------------------------------------------------------------------------------------
def _engine_f32(weights):
    """the same layer graphs on the exact f32-MFMA kernels (FCOSEngine / A2JEngine precision="f32")"""
    if weights not in _ENG32:
        from hn_amd.a2j_engine import A2JEngine
        from hn_amd.fcos_engine import FCOSEngine
        from hn_amd.pipeline import HandNetEngine
        _ENG32.clear()
        k = _classes(weights)
        _ENG32[weights] = HandNetEngine(FCOSEngine(_fcos_sd(weights[0]), k, device="cuda", precision="f32"),
                                        A2JEngine(_a2j_sd(weights[1]), device="cuda", precision="f32"), k)
    return _ENG32[weights]


def _compare(name, check_range, precision="f16x3"):
    import time
    from oracle import a2j_ref, fcos_ref, handnet_ref
    t_start = time.time()
    ora = _oracle(name)
    t_oracle = time.time()
    c = ora["case"]
    frames, depth = c["frames"], c["depth"]
    n = len(frames)
    f32 = precision == "f32"
    mode = "f32" if f32 else str(int(check_range))
    batch = torch.stack(frames).cuda()
    if f32:
        eng = _engine_f32(c["weights"])
        t_net = time.time()
        with torch.inference_mode():
            out = eng.forward_device(batch, depth.cuda())
            tup = tup2 = None
            s_hip = _hip_scores(eng, batch)
    else:
        net = _net(c["weights"])
        eng = net.engine()
        eng.check_range = bool(check_range)
        t_net = time.time()
        try:
            with torch.inference_mode():
                out = net.forward_device(batch, depth.cuda(), _graph=False)
                tup = net([f.cuda() for f in frames], depth_images=depth.cuda())
                tup2 = net([f.cuda() for f in frames], depth_images=depth.cuda())      # (second call: the sparse-stream path)
                s_hip = _hip_scores(eng, batch)
        finally:
            eng.check_range = False
    flips = _FLIPS.setdefault((name, mode), {})
    if out.range_flags is not None:
        assert out.range_flags.cpu().tolist() == [0, 0, 0, 0]
    t_hip = time.time()
    cnt = out.candidates.count.cpu().tolist()
    pts = out.candidates.point.cpu()
    cscores = out.candidates.scores.cpu()
    det = out.detections
    dcount = det.count.cpu().tolist()
    keep, labels, boxes = det.keep.cpu(), det.labels.cpu(), det.boxes.cpu()
    has = out.has_hand.cpu()
    box = out.crop_box.cpu()
    kp = out.keypoints.cpu()
    rgbd = _rgbd(c["weights"])
    crops = out.crops_nhwc.permute(0, 3, 1, 2).cpu() if rgbd else out.crops_nhwc[..., :1].permute(0, 3, 1, 2).cpu()   # [N, C, 176, 176]
    asd = _a2j_sd(c["weights"][1])
    identical, order_frames, tolerated, moved = 0, [], [], []   # moved: frames whose crop box differs for a tolerated reason
    deltas, worst_gap = [], 0.0
    for i in range(n):
        cand, rd = ora["cands"][i], ora["dets"][i]
        rp = cand["index"]
        hp, hk = pts[i, :cnt[i]].long(), keep[i, :dcount[i]].long()
        delta = float((s_hip[i].double() - ora["smax"][i].double()).abs().max())
        deltas.append(delta)
        assert delta <= SCORE_TOL, (name, i, delta)
        frame_name = (c.get("names") or [None] * n)[i]
        same_box = True
        if hp.tolist() != rp.tolist():
            ok, rec = _diagnose_candidates(name, i, ora, hp, s_hip[i].double(), delta)
            rec.update(check_range=bool(check_range), frame_name=frame_name)
            print(("TOLERATED (fp32-grade arithmetic does not determine this decision): " if ok else "MISMATCH: ") + json.dumps(rec))
            assert ok, rec
            tolerated.append(i)
            REPORT["tolerated"].append(rec)
            same_box = False        # (whatever follows works on another candidate list)
        else:
            hs = cscores[i, :cnt[i]]
            # bit-exact NMS: the reference algorithm on the oracle's boxes / labels in the HIP score order = the HIP survivors
            redo = fcos_ref.batched_nms(cand["boxes"], hs, cand["labels"], 0.3)
            assert redo.tolist() == hk.tolist(), (name, i, "NMS survivors differ from the reference NMS on the same score order")
            assert labels[i, :dcount[i]].long().tolist() == cand["labels"][hk].tolist(), (name, i)
            if hk.tolist() != rd["keep"].tolist():
                # the score ORDER differs: every pair the two sides rank differently must be closer than 2 delta
                os_ = cand["scores"].double()
                order_o = sorted(range(len(os_)), key=lambda k: (-float(os_[k]), k))
                rank_h = {k: r for r, k in enumerate(sorted(range(len(hs)), key=lambda k: (-float(hs[k]), k)))}
                seq = [rank_h[k] for k in order_o]
                inv, gap = 0, 0.0
                pairs = set()
                for a_ in range(len(seq)):
                    for b_ in range(a_ + 1, len(seq)):
                        if seq[a_] > seq[b_]:
                            inv += 1
                            gap = max(gap, abs(float(os_[order_o[a_]] - os_[order_o[b_]])))
                            pairs.add((order_o[a_], order_o[b_]))
                flips[i] = pairs
                ties = int((os_[order_o][:-1] == os_[order_o][1:]).sum()) if len(os_) > 1 else 0
                rec = {"case": name, "frame": i, "frame_name": frame_name, "kind": "score order of near-tied candidates",
                       "candidates": len(os_), "pairs_ranked_differently": inv, "largest_oracle_score_gap_of_such_a_pair": gap,
                       "exact_score_ties_in_the_oracle": ties, "hip_vs_oracle_score_delta": delta,
                       "survivor_set_identical": sorted(hk.tolist()) == sorted(rd["keep"].tolist()),
                       "check_range": bool(check_range), "precision": precision}
                assert inv > 0 and gap <= 2.0 * delta, rec
                worst_gap = max(worst_gap, gap)
                order_frames.append(i)
                REPORT["order"].append(rec)
            else:
                assert (boxes[i, :dcount[i]] - rd["boxes"]).abs().max().item() < 2e-2 if dcount[i] else True
        # ---- crop stage ----
        if bool(has[i]) != bool(ora["mask"][i]):
            same_box = False
        elif ora["mask"][i]:
            j = int(ora["mask"][:i].sum())
            same_box = same_box and box[i].tolist() == ora["boxes"][j].tolist()
        if not same_box and i not in tolerated and i not in order_frames:
            ok, rec = _diagnose_crop(name, i, ora, box[i], delta)
            rec.update(check_range=bool(check_range), frame_name=frame_name)
            print(("TOLERATED (fp32-grade arithmetic does not determine this decision): " if ok else "MISMATCH: ") + json.dumps(rec))
            assert ok, rec
            tolerated.append(i)
            REPORT["tolerated"].append(rec)
        if same_box:
            identical += int(i not in order_frames)
            if ora["mask"][i]:
                j = int(ora["mask"][:i].sum())
                assert torch.equal(crops[i], ora["dcrops"][j]), (name, i)       # pure gather (RGB-D: + the channel order): bit-exact
            else:
                assert box[i].tolist() == [0, 0, 0, 0] and float(kp[i].abs().max()) == 0.0
        else:
            moved.append(i)
    # ---- keypoints, every frame with a hand: the oracle's A2J on the oracle's crop, or (moved frames) on the HIP crop ----
    ref_kp = ora["kp"].clone()
    for i in moved:
        if has[i]:
            dc = handnet_ref.crop_depth(depth[i], box[i])
            dc = dc[[2, 1, 0, 3]] if (rgbd and dc is not None) else dc
            assert dc is not None and torch.equal(crops[i], dc), (name, i)
            ref_kp[i] = a2j_ref.a2j_forward(dc[None], asd, channel_in=4 if rgbd else 1)[0]
        else:
            ref_kp[i] = 0.0
    tol = max(KP_TOL, 3.0 * ora["noise64"])
    sel = has.bool()
    err = float((kp[sel] - ref_kp[sel]).abs().max()) if bool(sel.any()) else 0.0
    assert torch.isfinite(kp).all() and float(kp[~sel].abs().max() if bool((~sel).any()) else 0.0) == 0.0
    print(f"[parity sweep] {name} check_range={int(check_range)} precision={precision}: {n} frames, {int(ora['mask'].sum())} with a hand, candidates "
          f"{min(cnt)}..{max(cnt)}, survivors {min(dcount)}..{max(dcount)}; identical in every integer {identical}, order of "
          f"near-tied scores differs {order_frames} (largest gap {worst_gap:.1e}), tolerated decisions {tolerated}, crop box "
          f"moved {moved}; score delta {max(deltas):.1e}; max |dkp| {err:.2e} over {int(sel.sum())} frames (oracle fp32-vs-fp64 "
          f"on the same crops {ora['noise64']:.2e}, bound {tol:.1e}); seconds: oracle {t_oracle - t_start:.1f}, engines "
          f"{t_net - t_oracle:.1f}, HIP {t_hip - t_net:.1f}, comparison {time.time() - t_hip:.1f}")
    assert err < tol, (name, err, tol)
    REPORT["cases"][f"{name}/{mode}"] = {
        "precision": precision,
        "frames": n, "frames_with_hand": int(ora["mask"].sum()), "candidates": [min(cnt), max(cnt)],
        "survivors": [min(dcount), max(dcount)], "frames_identical_in_every_integer": identical,
        "frames_with_score_order_differences": order_frames, "tolerated_frames": tolerated, "crop_box_moved": moved,
        "max_score_delta": max(deltas), "max_abs_keypoint_diff": err, "oracle_fp32_vs_fp64": ora["noise64"], "bound": tol}
    # ---- the reference's return tuple through the drop-in callable (handnet_pipeline.py:107-116) ----
    if moved or f32:        # (the drop-in's constructor has no precision argument: the f32 pass is engine-level)
        return
    rkp, rdb, rcrops = ora["ref_tuple"]
    for t in (tup, tup2):
        gkp, gdb, gcrops = t
        assert gkp.device.type == "cpu" and gkp.shape == (n, 21, 3)
        assert (gkp - rkp).abs().max().item() < tol
        if bool(ora["mask"].any()):
            assert gcrops.dtype == torch.int64 and torch.equal(gcrops.cpu(), rcrops)
            assert torch.equal(gdb.cpu(), rdb)
        else:       # no frame with a hand: zeros, CPU float crops placeholder (:107-108)
            assert gcrops.dtype == torch.float32 and gcrops.shape == (n, 4) and float(gcrops.abs().max()) == 0.0
            assert gdb.shape == depth.shape and float(gdb.abs().max()) == 0.0


# case-major order: the oracle of a case is computed once and serves its two (F32_CASES: three) passes
_SWEEP = [(n, m) for n in CASES for m in ([0, 1] + (["f32"] if n in F32_CASES else []))]


@pytest.mark.parametrize("name,mode", _SWEEP, ids=[f"{n}-{m}" for n, m in _SWEEP])
def test_parity_sweep(name, mode):
    if mode == "f32":
        _compare(name, 0, precision="f32")
    else:
        _compare(name, mode)


def test_exact_f32_engines_and_the_near_tied_scores():
    """VERDICT r04 item 2c.  Runs after the sweep: for the cases that ran in both arithmetic modes, which frames rank near-tied
    scores differently from the oracle under the split-fp16 product (22-bit operands) and under the exact f32 MFMA (fp32
    operands, a k-ordered fmaf chain: only the SUMMATION ORDER differs from the oracle's CPU convolutions)?  Both modes passed
    every assertion of the sweep (identical candidate sets, bit-exact NMS given the score order, flipped pairs closer than twice
    the measured score difference); this test records the comparison and asserts the two things a sound diagnosis needs: the
    exact-f32 path flips near-ties too (so flipping is not a property of the 22-bit split), and no flipped pair of either mode
    is further apart than the oracle's own fp32-vs-fp64 noise allows."""
    rec = {}
    for name in F32_CASES:
        a, b = _FLIPS.get((name, "0")), _FLIPS.get((name, "f32"))
        if a is None or b is None:
            pytest.skip("the sweep cases of this comparison did not run in this session")
        fa, fb = set(a), set(b)
        both = sorted(fa & fb)
        jac = []
        for i in both:
            u = len(a[i] | b[i])
            jac.append(len(a[i] & b[i]) / u if u else 1.0)
        rec[name] = {"frames_flipped_f16x3": sorted(fa), "frames_flipped_f32": sorted(fb), "frames_flipped_in_both": both,
                     "pairs_f16x3": sum(len(v) for v in a.values()), "pairs_f32": sum(len(v) for v in b.values()),
                     "pair_overlap_jaccard_on_common_frames": [round(j, 3) for j in jac]}
    REPORT["f32_vs_f16x3_near_ties"] = rec
    print("[parity sweep] exact-f32 engines vs split-fp16, frames / pairs that rank near-tied scores differently from the oracle:",
          json.dumps(rec))
    total32 = sum(v["pairs_f32"] for v in rec.values())
    total16 = sum(v["pairs_f16x3"] for v in rec.values())
    # constant / saturated frames of `structured` hold EXACT ties in the oracle: any fp32-grade arithmetic orders them somehow
    assert total32 > 0 or total16 == 0, rec


def test_cases_reach_the_decision_points():
    """The sweep is only worth its name if the cases land where they aim (sized against the oracle in the build
    container; asserted here so that a change of the synthetic weights cannot silently defuse them)."""
    def facts(name):
        if name not in _FACTS:
            _oracle(name)
        return _FACTS[name]

    assert all(k > 1000 for k in facts("cands_1100")["candidates"])            # per-class branch of batched_nms
    assert all(k > 4096 for k in facts("cands_5200")["candidates"])
    m = facts("mixed_hands")["mask"]
    assert len(m) // 2 <= sum(m) < len(m)                                       # hand and no-hand frames in one batch
    m = facts("sparse_hands")["mask"]
    assert 0 < sum(m) < len(m) // 2                                             # ... few enough for the sparse-stream path
    assert any(k == 0 for k in facts("no_candidates")["candidates"])
    f = facts("structured")
    assert not f["mask"][f["names"].index("grey")] and f["mask"][f["names"].index("noise0")]


def test_tolerated_frames_are_rare():
    """Runs last in this module: every tolerated decision was printed with its offending values; they must stay the
    exception -- candidate-set / crop-box decisions on the threshold in <= 2 % of the frame evaluations, frames whose
    crop box moved (through those or through the order of tied scores) in <= 5 % -- and the report is written where
    gpurun brings it back."""
    frames = sum(v["frames"] for v in REPORT["cases"].values())
    moved = sum(len(v["crop_box_moved"]) for v in REPORT["cases"].values())
    order = sum(len(v["frames_with_score_order_differences"]) for v in REPORT["cases"].values())
    REPORT["summary"] = {"frame_evaluations": frames, "tolerated_decisions": len(REPORT["tolerated"]),
                         "frames_with_score_order_differences": order, "frames_whose_crop_box_moved": moved,
                         "frames_identical_in_every_integer": sum(v["frames_identical_in_every_integer"]
                                                                  for v in REPORT["cases"].values())}
    print("[parity sweep] summary:", json.dumps(REPORT["summary"]))
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "parity_sweep_report.json").write_text(json.dumps(REPORT, indent=1))
    except OSError:
        pass
    if frames:
        assert len(REPORT["tolerated"]) <= 0.02 * frames, REPORT["tolerated"]
        assert moved <= 0.05 * frames, REPORT["summary"]

"""Parity sweep: the drop-in HandNet on HIP against the oracle away from the single (seed-0 weights, white-noise frames)
operating point of the other end-to-end tests.

Cases (tests/parity_cases.py): other weight draws; structured frames (black / constant / saturated / ramps / checkerboard /
low-pass / one rectangle) mixed INTO one batch with noise frames; depth maps with zero holes and a constant depth map;
detectors whose output biases are raised until one image yields > 1000 and > 4000 candidates (torchvision's per-class
batched_nms branch, fcos_utils/fcos.py:635) and lowered until a batch mixes frames with and without a hand and frames with
no candidate at all.  Every case also runs with the range check on (HN_CHECK_RANGE=1's flag).

What is asserted per frame, against the oracle on the same inputs:
  candidate set (anchor-point indices that pass `scores_max > 0.7`, fcos.py:600) ......... identical
  NMS survivors (indices into the candidate list, score order) and labels ............... identical
  top-1 hand crop box (int64 truncation + 0.4 padding + clamp, handnet_pipeline.py:88-97) identical
  depth crops (pure gather) ............................................................. identical
  keypoints ............................................................................. |d| < max(1e-3, 3 x the
                                    oracle's own fp32-vs-fp64 difference on the same crops), both printed
A frame whose INTEGER results differ is never waved through: `_diagnose` re-runs the oracle for that frame in fp64 and the
frame is tolerated only if the deciding quantity (a score against 0.7, the gap of two scores the two sides RANK differently,
a box coordinate against an integer) sits inside 3 x max(the fp32 oracle's distance from its fp64 self, the measured HIP-vs-
oracle score difference of the frame, itself bounded by 5e-5) -- i.e. fp32-grade arithmetic does not determine the
decision: the reference's own CPU and CUDA runs would not agree on it either.  For a different survivor list it must also
hold that the REFERENCE NMS, fed the oracle's boxes in the HIP engine's score order, returns exactly the HIP survivors.
The offending values are printed and collected (gpurun_out/parity_sweep_report.json); the number of such frames is bounded
by `test_tolerated_frames_are_rare`.
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
REPORT = {"cases": {}, "tolerated": []}


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
    if name == "no_candidates":  # most frames without a single candidate
        return dict(weights=(("shift", -2.5), 0), frames=list(pc.noise_frames(8, 1600)), depth=pc.depth_noise(8, 2600))
    raise KeyError(name)


CASES = ["seed1", "seed2", "seed3", "structured", "depth_holes", "depth_constant", "cands_1100", "cands_5200",
         "mixed_hands", "sparse_hands", "no_candidates"]

_FCOS_SD, _A2J_SD, _NETS, _ORACLE, _FACTS = {}, {}, {}, {}, {}


def _fcos_sd(key):
    if key not in _FCOS_SD:
        from hn_amd import synth
        kind, v = key
        base = synth.make_fcos_state_dict(v if kind == "seed" else 0, NUM_CLASSES)
        _FCOS_SD[key] = base if kind == "seed" else pc.shift_detector_bias(base, cls_shift=v, num_classes=NUM_CLASSES)
    return _FCOS_SD[key]


def _a2j_sd(seed):
    if seed not in _A2J_SD:
        from hn_amd import synth
        _A2J_SD[seed] = synth.make_a2j_state_dict(seed)
    return _A2J_SD[seed]


def _net(weights):
    if weights not in _NETS:
        from handnet_pipeline.handnet_pipeline import HandNet
        args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j="unused.pth")
        net = HandNet(args, reload_detector=False, num_classes=NUM_CLASSES, reload_a2j=False, RGBD=False)
        net.detector.load_state_dict(_fcos_sd(weights[0]), strict=False)
        net.a2j.load_state_dict(_a2j_sd(weights[1]), strict=False)
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
    dets, cands = [], []
    for lo in range(0, len(frames), 4):
        d, inter = fcos_ref.fcos_forward(frames[lo:lo + 4], fsd, NUM_CLASSES, return_intermediates=True)
        dets += d
        cands += inter["candidates"]
    mask, boxes, dcrops = handnet_ref.select_and_crop(dets, depth, NUM_CLASSES)
    kp = torch.zeros((len(frames), 21, 3))
    noise64 = 0.0
    if dcrops:
        x = torch.stack(dcrops)
        k32 = a2j_ref.a2j_forward(x, asd)
        k64 = a2j_ref.a2j_forward(x, a2j_ref.to_dtype(asd, torch.float64), dtype=torch.float64)
        noise64 = float((k32.double() - k64).abs().max())
        kp[mask] = k32
    # the reference's return tuple (handnet_pipeline.py:107-116), assembled exactly as oracle/handnet_ref.handnet_forward does
    ref_tuple = (kp, torch.stack(dcrops), torch.stack(boxes)) if dcrops else \
        (kp, torch.zeros_like(depth), torch.zeros((len(frames), 4)))
    _FACTS[name] = dict(candidates=[len(x["index"]) for x in cands], mask=mask.tolist(), names=c.get("names"))
    _ORACLE.clear()
    _ORACLE[name] = dict(case=c, dets=dets, cands=cands, mask=mask, boxes=boxes, dcrops=dcrops, kp=kp, noise64=noise64,
                         ref_tuple=ref_tuple)
    return _ORACLE[name]


# ---------------------------------------------------------------------------------------------------------------------
# diagnosis of a frame whose integer results differ
# ---------------------------------------------------------------------------------------------------------------------
def _hip_scores(eng, frame):
    """scores_max of every anchor point from the HIP engine's own head tensors (fcos.py:593-598 on the device)"""
    cls_lr, reg_ctr, _, _ = eng.fcos.forward_heads(frame[None].cuda())
    cls = torch.cat([t.reshape(-1, t.shape[-1])[:, :NUM_CLASSES] for t in cls_lr])
    ctr = torch.cat([t.reshape(-1, t.shape[-1])[:, 4:5] for t in reg_ctr])
    return torch.sqrt(torch.sigmoid(cls) * torch.sigmoid(ctr)).max(dim=-1)[0].cpu()


def _iou(a, b):
    w = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    h = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = w * h
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _diagnose(name, i, eng, ora, hip_points, hip_keep, hip_box, hip_scores):
    """-> (tolerated?, record).  Re-runs the oracle for frame i in fp64 and looks for the decision the fp32 oracle
    itself does not determine.  hip_scores: the HIP engine's scores of its candidates (anchor order)."""
    from oracle import a2j_ref, fcos_ref, handnet_ref
    c = ora["case"]
    frame = c["frames"][i]
    fsd = _fcos_sd(c["weights"][0])
    _, i32 = fcos_ref.fcos_forward([frame], fsd, NUM_CLASSES, return_intermediates=True)
    d64, i64 = fcos_ref.fcos_forward([frame.double()], a2j_ref.to_dtype(fsd, torch.float64), NUM_CLASSES,
                                     return_intermediates=True)

    def smax(inter):
        ho = inter["head"]
        return torch.sqrt(torch.sigmoid(ho["cls_logits"][0]) * torch.sigmoid(ho["bbox_ctrness"][0])).max(dim=-1)[0].double()

    s32, s64, sh = smax(i32), smax(i64), _hip_scores(eng, frame).double()
    noise = float((s32 - s64).abs().max())                       # the fp32 oracle's own score noise on this frame
    hipdiff = float((sh - s32).abs().max())
    rec = {"case": name, "frame": i, "frame_name": (c.get("names") or [None] * (i + 1))[i],
           "oracle_fp32_vs_fp64_score_noise": noise, "hip_vs_oracle_fp32_score_diff": hipdiff}
    # the arithmetic contract on the scores themselves: whatever is tolerated below, the HIP scores stay this close
    if hipdiff > 5e-5:
        rec["kind"] = "scores differ by more than 5e-5"
        return False, rec
    margin = 3.0 * max(noise, hipdiff)
    rec["decision_margin"] = margin
    cand = ora["cands"][i]
    ref_points = cand["index"]
    sym = sorted(set(hip_points.tolist()) ^ set(ref_points.tolist()))
    if sym:
        rec["kind"] = "candidate set (score vs 0.7, fcos.py:600)"
        rec["points"] = [{"point": p, "oracle_fp32": float(s32[p]), "oracle_fp64": float(s64[p]), "hip": float(sh[p])}
                         for p in sym]
        worst = max(abs(float(s32[p]) - 0.7) for p in sym)
        rec["worst_margin_to_0.7"] = worst
        return worst <= margin, rec
    ref_keep = ora["dets"][i]["keep"]
    if hip_keep.tolist() != ref_keep.tolist():
        # Same candidates, different survivor list.  (1) the score ORDER: pairs of candidates the two sides rank differently
        # must be (near-)ties of the oracle; (2) given the HIP engine's order, the REFERENCE NMS on the oracle's boxes must
        # return exactly the HIP survivors -- then the order of (near-)tied scores is the whole difference.
        os_, hs = cand["scores"].double(), hip_scores.double()
        order_o = sorted(range(len(os_)), key=lambda k: (-float(os_[k]), k))
        order_h = sorted(range(len(hs)), key=lambda k: (-float(hs[k]), k))
        rank_h = {k: r for r, k in enumerate(order_h)}
        seq = [rank_h[k] for k in order_o]           # HIP ranks in the oracle's order: inversions = pairs ranked differently
        worst_gap, inversions, ties = 0.0, 0, int((os_[order_o][:-1] == os_[order_o][1:]).sum()) if len(os_) > 1 else 0
        for a_ in range(len(seq)):
            for b_ in range(a_ + 1, len(seq)):
                if seq[a_] > seq[b_]:
                    inversions += 1
                    worst_gap = max(worst_gap, abs(float(os_[order_o[a_]] - os_[order_o[b_]])))
        rec.update(kind="NMS survivors / score order (fcos.py:635)", candidates=len(os_), pairs_ranked_differently=inversions,
                   largest_oracle_score_gap_of_such_a_pair=worst_gap, exact_score_ties_in_the_oracle=ties)
        redo = fcos_ref.batched_nms(cand["boxes"], hip_scores.float(), cand["labels"], 0.3)
        rec["reference_nms_on_hip_score_order_gives_hip_survivors"] = redo.tolist() == hip_keep.tolist()
        if inversions and worst_gap <= margin and rec["reference_nms_on_hip_score_order_gives_hip_survivors"]:
            return True, rec
        # otherwise an IoU decided differently: report the margins of the candidates that changed sides
        cb, lab, sc = cand["boxes"], cand["labels"], cand["scores"]
        margins = []
        for j in sorted(set(hip_keep.tolist()) ^ set(ref_keep.tolist()))[:16]:
            best = None
            for k in range(len(sc)):
                if k != j and lab[k] == lab[j] and sc[k] >= sc[j]:
                    v = _iou(cb[k].tolist(), cb[j].tolist())
                    if best is None or abs(v - 0.3) < abs(best - 0.3):
                        best = v
            margins.append({"candidate": j, "closest_iou_to_0.3": best})
        rec["iou_margins"] = margins
        return False, rec
    # same survivors: the crop box differs through a coordinate that straddles an integer (handnet_pipeline.py:88-97)
    d = ora["dets"][i]
    hb = d["boxes"][d["labels"] == HAND][:1]
    hb64 = d64[0]["boxes"][d64[0]["labels"] == HAND][:1]
    rec["kind"] = "crop box (int64 truncation, handnet_pipeline.py:88)"
    rec["oracle_fp32_top_hand_box"] = hb.tolist()
    rec["oracle_fp64_top_hand_box"] = hb64.tolist()
    rec["hip_crop"] = hip_box.tolist()
    rec["oracle_crop"] = handnet_ref.crop_box(hb, pc.W, pc.H).tolist() if len(hb) else None
    if not len(hb) or not len(hb64):
        return False, rec
    coord_noise = float((hb.double() - hb64).abs().max())
    dist = float((hb - hb.round()).abs().min())
    rec.update(oracle_fp32_vs_fp64_coordinate_noise=coord_noise, min_distance_to_integer=dist)
    return dist <= 3.0 * coord_noise, rec


# ---------------------------------------------------------------------------------------------------------------------
# the sweep
# ---------------------------------------------------------------------------------------------------------------------
def _compare(name, check_range):
    ora = _oracle(name)
    c = ora["case"]
    frames, depth = c["frames"], c["depth"]
    n = len(frames)
    net = _net(c["weights"])
    eng = net.engine()
    eng.check_range = bool(check_range)
    try:
        with torch.inference_mode():
            out = net.forward_device(torch.stack(frames).cuda(), depth.cuda(), _graph=False)
            tup = net([f.cuda() for f in frames], depth_images=depth.cuda())
            tup2 = net([f.cuda() for f in frames], depth_images=depth.cuda())      # (second call: the sparse-stream path)
    finally:
        eng.check_range = False
    cnt = out.candidates.count.cpu().tolist()
    pts = out.candidates.point.cpu()
    cscores = out.candidates.scores.cpu()
    det = out.detections
    dcount = det.count.cpu().tolist()
    keep, labels, boxes = det.keep.cpu(), det.labels.cpu(), det.boxes.cpu()
    has = out.has_hand.bool().cpu()
    box = out.crop_box.cpu()
    kp = out.keypoints.cpu()
    crops = out.crops_nhwc[..., 0].cpu()
    tolerated = []
    for i in range(n):
        rp, rd = ora["cands"][i]["index"], ora["dets"][i]
        hp, hk = pts[i, :cnt[i]].long(), keep[i, :dcount[i]].long()
        integer_ok = (hp.tolist() == rp.tolist() and hk.tolist() == rd["keep"].tolist()
                      and labels[i, :dcount[i]].long().tolist() == rd["labels"].tolist()
                      and bool(has[i]) == bool(ora["mask"][i]))
        if integer_ok and ora["mask"][i]:
            j = int(ora["mask"][:i].sum())
            integer_ok = box[i].tolist() == ora["boxes"][j].tolist()
        if not integer_ok:
            ok, rec = _diagnose(name, i, eng, ora, hp, hk, box[i], cscores[i, :cnt[i]])
            rec["check_range"] = bool(check_range)
            print(("TOLERATED (the fp32 oracle does not determine this decision either): " if ok else "MISMATCH: ")
                  + json.dumps(rec))
            assert ok, rec
            tolerated.append(i)
            REPORT["tolerated"].append(rec)
            continue
        # detection boxes / scores: fp32 values computed from logits that differ in the last bits
        if dcount[i]:
            assert (boxes[i, :dcount[i]] - rd["boxes"]).abs().max().item() < 2e-2, (name, i)
        if ora["mask"][i]:
            j = int(ora["mask"][:i].sum())
            assert torch.equal(crops[i], ora["dcrops"][j][0]), (name, i)       # pure gather: bit-exact
        else:
            assert box[i].tolist() == [0, 0, 0, 0] and float(kp[i].abs().max()) == 0.0
    good = [i for i in range(n) if i not in tolerated and bool(ora["mask"][i])]
    tol = max(KP_TOL, 3.0 * ora["noise64"])
    err = float((kp[good] - ora["kp"][good]).abs().max()) if good else 0.0
    assert torch.isfinite(kp).all()
    print(f"[parity sweep] {name} check_range={int(check_range)}: {n} frames, {int(ora['mask'].sum())} with a hand, candidates "
          f"{min(cnt)}..{max(cnt)}, survivors {min(dcount)}..{max(dcount)}, max |dkp| {err:.2e} (oracle fp32-vs-fp64 on the "
          f"same crops {ora['noise64']:.2e}, bound {tol:.1e}), tolerated frames {tolerated}")
    assert err < tol, (name, err, tol)
    REPORT["cases"][f"{name}/{int(check_range)}"] = {
        "frames": n, "frames_with_hand": int(ora["mask"].sum()), "candidates": [min(cnt), max(cnt)],
        "survivors": [min(dcount), max(dcount)], "max_abs_keypoint_diff": err, "oracle_fp32_vs_fp64": ora["noise64"],
        "bound": tol, "tolerated_frames": tolerated}
    # ---- the reference's return tuple through the drop-in callable (handnet_pipeline.py:107-116) ----
    if tolerated:
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


@pytest.mark.parametrize("check_range", [0, 1])
@pytest.mark.parametrize("name", CASES)
def test_parity_sweep(name, check_range):
    _compare(name, check_range)


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
    """Runs last in this module: every tolerated frame was printed with its offending values; they must stay the
    exception (<= 2 % of the frame evaluations) -- and the report is written where gpurun brings it back."""
    frames = sum(v["frames"] for v in REPORT["cases"].values())
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "parity_sweep_report.json").write_text(json.dumps(REPORT, indent=1))
    except OSError:
        pass
    if frames:
        assert len(REPORT["tolerated"]) <= 0.02 * frames, REPORT["tolerated"]

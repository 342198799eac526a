"""HIP engine vs oracle on whole frames: where end-to-end parity can actually break.

Test infrastructure only (imported by tests/ and by bench.py's cpu_baseline leg, never by the product).
A 1-ulp change of a score across 0.7 (fcos_utils/fcos.py:600), of an IoU across 0.3 (:635) or of a box
coordinate across an integer (handnet_pipeline.py:88 `.to(int64)`) changes the integer crop and moves
the keypoints by O(1 px) -- 1000x the 1e-3 tolerance.  So agreement is reported as: the rate of frames
whose integer crop box is identical, the keypoint difference on those frames, and for every other frame
WHY it flipped (margins of the quantities that decided it).
"""
from __future__ import annotations

import time

import torch

from . import a2j_ref, fcos_ref, handnet_ref


def _top_hand(boxes, scores, labels, hand_label):
    """first hand-class detection of a score-ordered list -> (box [4] fp32, score, rank) or None"""
    idx = torch.where(labels == hand_label)[0]
    if idx.numel() == 0:
        return None
    i = int(idx[0])
    return boxes[i].float(), float(scores[i]), i


def pipeline_parity(engine, rgb, depth, fcos_sd, a2j_sd, num_classes=3, chunk=8, tolerance=1e-3):
    """rgb [N,3,H,W], depth [N,1,H,W] CPU tensors; engine: hn_amd.pipeline.HandNetEngine on the GPU.
    Returns (stats dict, oracle seconds).  The oracle runs in chunks of `chunk` frames."""
    n = rgb.shape[0]
    dev = engine.fcos.device
    hand = num_classes - 1
    H, W = depth.shape[-2:]
    # ---- oracle ----
    t0 = time.time()
    o_kp = torch.zeros((n, 21, 3))
    o_box = torch.zeros((n, 4), dtype=torch.int64)
    o_has = torch.zeros((n,), dtype=torch.bool)
    o_dets = []
    with torch.no_grad():
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            dets = fcos_ref.fcos_forward([rgb[i] for i in range(lo, hi)], fcos_sd, num_classes)
            o_dets += dets
            mask, boxes, dcrops = handnet_ref.select_and_crop(dets, depth[lo:hi], num_classes)
            if dcrops:
                kp = a2j_ref.a2j_forward(torch.stack(dcrops), a2j_sd)
                sel = torch.where(mask)[0] + lo
                o_kp[sel] = kp
                o_box[sel] = torch.stack(boxes)
                o_has[sel] = True
    oracle_s = time.time() - t0
    # ---- HIP engine (same chunking does not matter: frames are independent; one batch per chunk of 32) ----
    g_kp, g_box, g_has, g_det = [], [], [], []
    for lo in range(0, n, 32):
        hi = min(n, lo + 32)
        out = engine.forward_device(rgb[lo:hi].to(dev), depth[lo:hi].to(dev))
        g_kp.append(out.keypoints.cpu())
        g_box.append(out.crop_box.cpu())
        g_has.append(out.has_hand.bool().cpu())
        d = out.detections
        cnt = d.count.cpu().tolist()
        bx, sc, lb = d.boxes.cpu(), d.scores.cpu(), d.labels.cpu()
        g_det += [(bx[i, :k], sc[i, :k], lb[i, :k]) for i, k in enumerate(cnt)]
    g_kp, g_box, g_has = torch.cat(g_kp), torch.cat(g_box), torch.cat(g_has)
    # ---- compare ----
    same = (g_has == o_has) & ((g_box == o_box).all(dim=1) | ~o_has)
    flips = []
    for i in torch.where(~same)[0].tolist():
        od = o_dets[i]
        o_top = _top_hand(od["boxes"], od["scores"], od["labels"], hand)
        g_top = _top_hand(*g_det[i], hand)
        rec = {"frame": i, "oracle_crop": o_box[i].tolist(), "hip_crop": g_box[i].tolist(),
               "oracle_detections": int(od["scores"].numel()), "hip_detections": int(g_det[i][1].numel())}
        if o_top is None or g_top is None:
            rec["cause"] = "hand present on one side only (score vs 0.7 or NMS decision)"
        else:
            dbox = float((o_top[0] - g_top[0]).abs().max())
            rec.update(top_hand_score=[o_top[1], g_top[1]], top_hand_box_max_abs_diff=dbox,
                       # distance of the oracle's float coordinates (and of the padded ones) to the next integer:
                       # a coordinate this close to an integer truncates differently under a 1e-4 px perturbation
                       min_dist_to_integer=float(((o_top[0] - o_top[0].round()).abs()).min()))
            if dbox < 0.05:
                rec["cause"] = "same detection, a coordinate straddles an integer (int64 truncation / 0.4 padding)"
            else:
                sc = od["scores"][od["labels"] == hand]
                rec["cause"] = "different top-1 hand detection (ranking / NMS)"
                rec["oracle_top2_score_gap"] = float(sc[0] - sc[1]) if sc.numel() > 1 else None
        flips.append(rec)
    eq = same & o_has
    kp_diff = float((g_kp[eq] - o_kp[eq]).abs().max()) if bool(eq.any()) else 0.0
    kp_mean = float((g_kp[eq] - o_kp[eq]).abs().mean()) if bool(eq.any()) else 0.0
    stats = {"frames": n, "frames_with_hand": int(o_has.sum()), "crop_box_equal_frames": int(same.sum()),
             "crop_box_equality_rate": round(float(same.float().mean()), 6),
             "crop_boxes_identical": bool(same.all()),
             "max_abs_keypoint_diff": kp_diff, "mean_abs_keypoint_diff": kp_mean, "tolerance": tolerance,
             "keypoints_within_tolerance": bool(kp_diff < tolerance), "flips": flips}
    return stats, oracle_s, (g_kp, g_box, g_has, o_kp, o_box, o_has)

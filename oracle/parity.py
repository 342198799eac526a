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
    Returns (stats dict, oracle seconds, (HIP keypoints / boxes / flags, oracle keypoints / boxes / flags, the oracle's
    detection dicts per frame)).  The oracle runs in chunks of `chunk` frames."""
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
    return stats, oracle_s, (g_kp, g_box, g_has, o_kp, o_box, o_has, o_dets)


def _iou(a, b):
    """row-wise IoU of two [k,4] (x1,y1,x2,y2) box lists, in float64"""
    a, b = a.double(), b.double()
    ix = (torch.minimum(a[:, 2], b[:, 2]) - torch.maximum(a[:, 0], b[:, 0])).clamp(min=0)
    iy = (torch.minimum(a[:, 3], b[:, 3]) - torch.maximum(a[:, 1], b[:, 1])).clamp(min=0)
    inter = ix * iy
    union = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
    return torch.where(union > 0, inter / union, torch.ones_like(union))


def fcos_parity(hip_dets, oracle_dets):
    """Detector agreement in the terms BASELINE.json config 3 names ("box IoU vs torchvision ref", SURVEY 8d: "+ survivor-index
    equality"), per frame and over the batch.
      hip_dets     per frame (boxes [k,4], scores [k], labels [k], keep [k]) CPU tensors from the HIP detector; keep = index of
                   each survivor in the frame's anchor-ordered candidate list (fcos_utils/fcos.py:624-635), score-descending
      oracle_dets  oracle.fcos_ref.fcos_forward's dicts for the same frames (they carry the same `keep`)
    Survivors are matched BY THAT INDEX (the same anchor point); IoU / score / label figures are over matched survivors."""
    frames = len(hip_dets)
    list_equal = set_equal = 0
    ious, dscore, label_eq, matched, total_o, total_h = [], 0.0, 0, 0, 0, 0
    for (hb, hs, hl, hk), od in zip(hip_dets, oracle_dets):
        hk = hk.to(torch.int64)
        ok = od["keep"].to(torch.int64)
        total_o += int(ok.numel())
        total_h += int(hk.numel())
        list_equal += int(hk.numel() == ok.numel() and bool((hk == ok).all()))
        set_equal += int(set(hk.tolist()) == set(ok.tolist()))
        pos = {int(v): i for i, v in enumerate(ok.tolist())}
        hi = [i for i, v in enumerate(hk.tolist()) if int(v) in pos]
        if not hi:
            continue
        oi = [pos[int(hk[i])] for i in hi]
        hi, oi = torch.tensor(hi), torch.tensor(oi)
        ious.append(_iou(hb[hi], od["boxes"][oi]))
        dscore = max(dscore, float((hs[hi] - od["scores"][oi]).abs().max()))
        label_eq += int((hl[hi].to(torch.int64) == od["labels"][oi].to(torch.int64)).sum())
        matched += int(hi.numel())
    iou = torch.cat(ious) if ious else torch.ones(0, dtype=torch.float64)
    return {"frames": frames, "survivors_oracle": total_o, "survivors_hip": total_h, "matched_survivors": matched,
            "survivor_index_list_equal_frames": list_equal, "survivor_index_list_equality_rate": round(list_equal / max(1, frames), 4),
            "survivor_index_set_equal_frames": set_equal,
            "label_equality_rate": round(label_eq / max(1, matched), 6),
            "mean_box_iou": round(float(iou.mean()), 9) if iou.numel() else None,
            "min_box_iou": round(float(iou.min()), 9) if iou.numel() else None,
            "max_abs_score_diff": dscore,
            "what": "HIP detector vs oracle.fcos_ref.fcos_forward on the same frames; survivors matched by their index in the "
                    "anchor-ordered candidate list; a frame counts as list-equal when the score-ordered survivor indices are "
                    "identical (near-tied scores may exchange places: DESIGN.md section 5)"}


def a2j_parity(hip_kp, crops, a2j_sd, box=(224, 152, 400, 328), paras=(617.343, 617.343, 312.42, 241.42), tolerance=1e-3):
    """A2J-only agreement in the terms BASELINE.json config 2 names ("EPE vs CPU ref"): hip_kp [k,21,3] (CPU) from the HIP
    engine for crops [k,1,176,176] vs oracle.a2j_ref.a2j_forward on the same crops -- max / mean |d(u,v,d)| in crop units
    and the end-point error in millimetres after convert_joints + uvd2xyz (a2j/a2j.py:17-43, a2jdataset.py:31-38) with ONE
    nominal 176 x 176 crop box around the principal point (an A2J-only run has no detector boxes; SURVEY 8d's intrinsics)."""
    import numpy as np
    t0 = time.time()
    o_kp = a2j_ref.a2j_forward(crops, a2j_sd)
    oracle_s = time.time() - t0
    d = (hip_kp - o_kp).abs()
    epe = []
    for i in range(hip_kp.shape[0]):
        a = a2j_ref.convert_joints(hip_kp[i].numpy(), np.asarray(box), paras)
        b = a2j_ref.convert_joints(o_kp[i].numpy(), np.asarray(box), paras)
        epe.append(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=1).mean())
    return {"crops": int(hip_kp.shape[0]), "max_abs_uvd_diff": float(d.max()), "mean_abs_uvd_diff": float(d.mean()),
            "mm_epe": float(np.mean(epe)), "max_crop_mm_epe": float(np.max(epe)), "tolerance": tolerance,
            "keypoints_within_tolerance": bool(float(d.max()) < tolerance), "crop_box_for_mm": list(box), "paras": list(paras),
            "what": "HIP A2J (rows of the batch-64 step) vs oracle.a2j_ref.a2j_forward (torch CPU fp32) on the same crops"}, oracle_s


def live_parity(hip, rgb, depth, fcos_sd, a2j_sd, p2m_sd, graphs, paras, num_classes=3, reps=5):
    """The live caller's chain on the frames of bench.py's `live_b1` step (ros_demo.py:270-290,329-337,148-161) through the
    oracle: handnet_ref -> the caller's clamps -> a2j_ref.convert_joints (twice: image uv, camera xyz) -> pose2mesh_ref.lifter_input
    -> pose2mesh_ref.pose2mesh_forward, against what the captured HIP step copied to the host:
      hip = (keypoints [n,21,3], crop_box [n,4], image_uvd [n,21,3], xyz_mm [n,21,3], mesh [n,V0,3]) CPU tensors.
    Returns (stats, median seconds of `reps` timed oracle chains per frame after one warm-up, median seconds of the lifter part
    alone).  The mesh figure is on frames whose integer crop box is identical (else the keypoints differ by O(1 px), see above)."""
    import numpy as np
    from . import pose2mesh_ref
    h_kp, h_box, h_img, h_xyz, h_mesh = hip
    n = h_kp.shape[0]
    H, W = depth.shape[-2:]

    def chain(i):
        t0 = time.time()
        o_kp, _d, o_crops = handnet_ref.handnet_forward([rgb[i]], depth[i:i + 1], fcos_sd, a2j_sd, num_classes)
        det = o_crops[0].clone()
        det[:2] = torch.clamp(det[:2], 0, H)              # ros_demo.py:281-283
        det[2:] = torch.clamp(det[2:], 0, W)
        k = torch.clamp(o_kp[0], min=0.0, max=176.0).numpy()
        j2d = a2j_ref.convert_joints(k, det.numpy(), None)
        j3d = a2j_ref.convert_joints(k, det.numpy(), paras)
        t1 = time.time()
        x = pose2mesh_ref.lifter_input(j2d[:, :2])
        mesh = None
        if x is not None:
            mesh, _pose3d = pose2mesh_ref.pose2mesh_forward(torch.from_numpy(x)[None], p2m_sd, graphs)
        t2 = time.time()
        return (o_kp[0], o_crops[0], j2d, j3d, mesh), t2 - t0, t2 - t1

    chain(0)
    whole, lift = [], []
    same_box, d_kp, d_img, d_xyz, d_mesh = 0, 0.0, 0.0, 0.0, 0.0
    for i in range(n):
        for r in range(reps if i == 0 else 1):
            (o_kp, o_box, j2d, j3d, o_mesh), s, sl = chain(i)
            if i == 0:
                whole.append(s)
                lift.append(sl)
        if not torch.equal(o_box, h_box[i]) or o_mesh is None:
            continue
        same_box += 1
        d_kp = max(d_kp, float((o_kp - h_kp[i]).abs().max()))
        d_img = max(d_img, float(np.abs(h_img[i].numpy()[:, :2] - j2d[:, :2]).max()))
        d_xyz = max(d_xyz, float(np.abs(h_xyz[i].numpy() - j3d).max()))
        d_mesh = max(d_mesh, float((o_mesh[0] - h_mesh[i]).abs().max()))
    stats = {"frames": int(n), "crop_box_identical": int(same_box), "max_abs_keypoint_diff": d_kp, "max_abs_image_uv_diff_px": d_img,
             "max_abs_xyz_diff_mm": d_xyz, "max_abs_mesh_vertex_diff": d_mesh, "mesh_tolerance": 2e-3,
             "mesh_within_tolerance": bool(same_box > 0 and d_mesh < 2e-3),
             "what": "the captured live step's host record (keypoints, image uv, camera xyz, Pose2Mesh vertices) vs the oracle's "
                     "chain on the same frame: handnet_ref -> clamps -> a2j_ref.convert_joints -> pose2mesh_ref.lifter_input -> "
                     "pose2mesh_ref.pose2mesh_forward (torch CPU fp32 + numpy glue)"}
    return stats, sorted(whole)[len(whole) // 2], sorted(lift)[len(lift) // 2]

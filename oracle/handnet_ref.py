"""Oracle for the end-to-end HandNet glue (handnet_pipeline/handnet_pipeline.py:58-116).
Test infrastructure only.  Pinned by tests/golden/handnet_forward.npz and handnet_rgbd_forward.npz.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import a2j_ref, fcos_ref


def crop_box(box, width, height, percent=0.4):
    """handnet_pipeline.py:88-97: fp32 box -> padded, clamped int64 box (x1,y1,x2,y2)."""
    box = box.reshape(4).to(torch.int64)
    w = box[2] - box[0]
    h = box[3] - box[1]
    box = box.clone()
    box[0] = max(0, box[0] - percent * (w))
    box[1] = max(0, box[1] - percent * (h))
    box[2] = min(width, box[2] + percent * (w))
    box[3] = min(height, box[3] + percent * (h))
    return box


def crop_depth(depth_img, box, size=176):
    """handnet_pipeline.py:101: inclusive slice + nearest resize.  depth_img [C,H,W]."""
    sl = depth_img[:, box[1]:box[3] + 1, box[0]:box[2] + 1]
    if sl.shape[1] == 0 or sl.shape[2] == 0:
        return None
    return F.interpolate(sl.unsqueeze(0), size=(size, size)).squeeze(0)


def select_and_crop(dets, depth_images, num_classes, rgbd=False):
    """handnet_pipeline.py:74-105.  Returns (image_mask, crops list, depth crops list).

    Deviation from the reference (documented in DESIGN.md): a frame without a hand-class
    detection, or whose padded box gives an empty slice, is reported as `no hand`
    (mask False) instead of raising / reusing the previous frame's crop.
    """
    n, _, H, W = depth_images.shape
    mask = torch.zeros(n, dtype=torch.bool)
    boxes, dcrops = [], []
    for i, d in enumerate(dets):
        hb = d["boxes"][d["labels"] == num_classes - 1]
        if len(hb) == 0:
            continue
        box = crop_box(hb[:1], W, H)
        dc = crop_depth(depth_images[i], box)
        if dc is None:
            continue
        if rgbd:
            dc = dc[[2, 1, 0, 3], :, :]  # handnet_pipeline.py:102
        mask[i] = True
        boxes.append(box)
        dcrops.append(dc)
    return mask, boxes, dcrops


def handnet_forward(images, depth_images, fcos_sd, a2j_sd, num_classes=3, rgbd=False):
    """images: list of [3,H,W]; depth_images [N,1,H,W] (rgbd: [N,4,H,W]) -> (keypoints [N,21,3], depth_batch, crops)."""
    with torch.no_grad():
        n = len(images)
        final = torch.zeros((n, 21, 3))
        dets = fcos_ref.fcos_forward(images, fcos_sd, num_classes)
        mask, boxes, dcrops = select_and_crop(dets, depth_images, num_classes, rgbd)
        if not dcrops:
            return final, torch.zeros_like(depth_images), torch.zeros((n, 4))
        depth_batch = torch.stack(dcrops)
        crops = torch.stack(boxes)
        final[mask] = a2j_ref.a2j_forward(depth_batch, a2j_sd, channel_in=4 if rgbd else 1)
    return final, depth_batch, crops

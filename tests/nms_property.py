"""Definition-level anchor for greedy NMS (not a test module; shared by the CPU and GPU tests).

For tie-free scores, greedy NMS is the UNIQUE subset K of the boxes such that, within every class,
  (i)  no two kept boxes overlap by more than the threshold, and
  (ii) every suppressed box overlaps a HIGHER-scored kept box by more than the threshold
(induction over the score order: the top box is kept by (ii); each next box is decided by the kept boxes above it).
`check` verifies (i) and (ii) with IoUs computed independently in fp64; `make_case` draws boxes and rejects draws in
which any same-class IoU falls inside a guard band around the threshold, so fp32 / fp64 rounding of the IoU cannot
flip a decision.  This pins hn_nms / hn_fcos_nms and oracle/nms_ref.c on the definition of torchvision.ops.nms
(torchvision 0.11.3 torchvision/csrc/ops/cpu/nms_kernel.cpp), independently of each other."""
import numpy as np
import torch


def iou64(boxes: torch.Tensor) -> np.ndarray:
    b = boxes.double().numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    x1 = np.maximum(b[:, None, 0], b[None, :, 0]); y1 = np.maximum(b[:, None, 1], b[None, :, 1])
    x2 = np.minimum(b[:, None, 2], b[None, :, 2]); y2 = np.minimum(b[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    return inter / (area[:, None] + area[None, :] - inter)


def make_case(k, classes, seed, thr=0.3, band=1e-4, extent=600.0):
    """k clustered boxes (plenty of overlaps on both sides of the threshold) with distinct scores; boxes taking part in a
    same-class pair whose fp64 IoU lies within `band` of the threshold are dropped (the draw is 25 % larger to make up for
    it), so that no decision hinges on the rounding of an IoU."""
    g = torch.Generator().manual_seed(seed)
    n = k + k // 4 + 8
    centres = torch.rand((max(4, n // 8), 2), generator=g) * extent
    c = centres[torch.randint(0, centres.shape[0], (n,), generator=g)] + torch.randn((n, 2), generator=g) * 12.0
    wh = 20.0 + torch.rand((n, 2), generator=g) * 60.0
    boxes = torch.cat([c - wh / 2, c + wh / 2], dim=1).float().contiguous()
    scores = torch.rand((n,), generator=g).float()
    labels = torch.randint(0, classes, (n,), generator=g, dtype=torch.int32)
    iou = iou64(boxes)
    same = labels[:, None].numpy() == labels[None, :].numpy()
    np.fill_diagonal(same, False)
    near = same & (np.abs(iou - thr) < band)
    drop = np.zeros((n,), dtype=bool)
    for a, b in np.argwhere(np.triu(near)):
        if not drop[a] and not drop[b]:
            drop[b] = True
    # distinct scores: drop later duplicates
    _, first = np.unique(scores.numpy(), return_index=True)
    dup = np.ones((n,), dtype=bool)
    dup[first] = False
    keep = torch.from_numpy(np.where(~(drop | dup))[0][:k])
    assert keep.numel() == k, (keep.numel(), k)
    return boxes[keep].contiguous(), scores[keep].contiguous(), labels[keep].contiguous()


def check(boxes, scores, labels, keep, thr=0.3):
    """keep: int64 indices of the kept boxes.  Raises AssertionError with the offending pair."""
    k = boxes.shape[0]
    keep = np.asarray(keep, dtype=np.int64)
    assert len(set(keep.tolist())) == len(keep) and (len(keep) == 0 or (keep.min() >= 0 and keep.max() < k))
    iou = iou64(boxes)
    lab = labels.numpy()
    sc = scores.double().numpy()
    kept = np.zeros((k,), dtype=bool)
    kept[keep] = True
    same = lab[:, None] == lab[None, :]
    over = (iou > thr) & same
    np.fill_diagonal(over, False)
    # (i) kept boxes of one class do not overlap beyond the threshold
    bad = np.argwhere(over & kept[:, None] & kept[None, :])
    assert bad.size == 0, f"kept boxes {bad[0].tolist()} overlap by {iou[tuple(bad[0])]:.6f} > {thr}"
    # (ii) every suppressed box has a higher-scored kept box of its class overlapping it beyond the threshold
    higher = sc[None, :] > sc[:, None]                       # [i, j]: j scores higher than i
    covered = (over & kept[None, :] & higher).any(axis=1)
    missing = np.where(~kept & ~covered)[0]
    assert missing.size == 0, f"box {missing[0]} was suppressed without a higher-scored overlapping kept box"
    # and the kept list is in descending score order (torchvision returns it that way)
    assert np.all(np.diff(sc[keep]) < 0)

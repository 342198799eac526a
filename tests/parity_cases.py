"""Inputs of the parity sweep (tests/test_parity_sweep_gpu.py): structured frames, depth maps with holes, and detector
checkpoints pushed towards the reference's decision points.  Test infrastructure only; every tensor is a pure function of
its arguments, so this container (where the cases were sized against the oracle) and the GPU box see the same bytes.

Decision points of the reference the cases aim at:
  scores_max > 0.7 ................................ fcos_utils/fcos.py:600
  batched_nms(.., 0.3), per-class branch K > 1000 .. fcos_utils/fcos.py:635 (torchvision boxes.py: numel() > 4000)
  int64 truncation + 0.4 padding + clamp .......... handnet_pipeline/handnet_pipeline.py:88-97
  GroupNorm over near-constant maps ............... fcos_utils/fcos.py:232-239 (rsqrt(var + eps) with var -> 0)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

H, W = 480, 640


def _noise(seed, n=1, h=H, w=W):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((n, 3, h, w), generator=g, dtype=torch.float32)


def structured_frames(h: int = H, w: int = W) -> "dict[str, torch.Tensor]":
    """name -> [3,h,w] fp32 in 0..1"""
    xs = torch.arange(w, dtype=torch.float32) / float(w - 1)
    ys = torch.arange(h, dtype=torch.float32) / float(h - 1)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    checker = (((yy // 8) + (xx // 8)) % 2).to(torch.float32)
    # low-pass ("natural-like") noise: white noise on a 15 x 20 grid, bicubically enlarged, clamped to 0..1
    coarse = _noise(77, 1, h // 32, w // 32)
    smooth = F.interpolate(coarse, size=(h, w), mode="bicubic", align_corners=False).clamp_(0.0, 1.0)[0]
    # a bright rectangle on a dark ground: one sharp object, flat elsewhere
    rect = torch.full((3, h, w), 0.1)
    rect[:, 140:330, 220:430] = 0.9
    return {
        "black": torch.zeros((3, h, w)),
        "grey": torch.full((3, h, w), 0.5),
        "white": torch.ones((3, h, w)),
        "ramp": xs.reshape(1, 1, w).expand(3, h, w).contiguous(),
        "ramp_v": torch.stack([ys.reshape(h, 1).expand(h, w), 1.0 - ys.reshape(h, 1).expand(h, w),
                               xs.reshape(1, w).expand(h, w)]).contiguous(),
        "checker8": checker.reshape(1, h, w).expand(3, h, w).contiguous(),
        "lowpass": smooth.contiguous(),
        "rect": rect,
    }


def noise_frames(n: int, seed: int = 1000, h: int = H, w: int = W) -> torch.Tensor:
    return _noise(seed, n, h, w)


def depth_noise(n: int, seed: int = 2000, h: int = H, w: int = W) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return 0.3 + 1.2 * torch.rand((n, 1, h, w), generator=g, dtype=torch.float32)


def depth_with_holes(n: int, seed: int = 2100, h: int = H, w: int = W) -> torch.Tensor:
    """Sensor-like depth: a smooth surface in metres with zero-valued holes (invalid pixels of a depth camera) -- square
    patches and salt noise -- covering about a fifth of the map."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand((n, 1, h // 40, w // 40), generator=g, dtype=torch.float32)
    d = 0.4 + 0.9 * F.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False)
    salt = torch.rand((n, 1, h, w), generator=g) < 0.08
    d = torch.where(salt, torch.zeros(()), d)
    for i in range(n):
        for _ in range(12):
            y = int(torch.randint(0, h - 40, (1,), generator=g))
            x = int(torch.randint(0, w - 40, (1,), generator=g))
            s = int(torch.randint(8, 40, (1,), generator=g))
            d[i, :, y:y + s, x:x + s] = 0.0
    return d.contiguous()


def depth_constant(n: int, value: float = 0.75, h: int = H, w: int = W) -> torch.Tensor:
    return torch.full((n, 1, h, w), float(value), dtype=torch.float32)


def shift_detector_bias(fcos_sd, cls_shift: float = 0.0, ctr_shift: float = 0.0, hand_shift: float = 0.0, num_classes=3):
    """A copy of a FCOS checkpoint with the cls_logits / bbox_ctrness biases moved: raises (or lowers) every score, i.e.
    the number of points that pass `scores_max > 0.7`; hand_shift moves the hand class (label num_classes - 1) alone."""
    sd = {k: v.clone() for k, v in fcos_sd.items()}
    b = sd["head.classification_head.cls_logits.bias"]
    b += cls_shift
    b[num_classes - 1] += hand_shift
    sd["head.regression_head.bbox_ctrness.bias"] += ctr_shift
    return sd

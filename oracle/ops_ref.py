"""Per-op CPU references (plain torch fp32) for the HIP kernels.  Test infrastructure only.

Each function mirrors one C-ABI entry point of include/handnet_hip.h with torch ops
that the reference itself calls (F.conv2d, F.max_pool2d, F.group_norm, F.interpolate).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def conv2d_nhwc(x, w, bias=None, stride=1, pad=0, dil=1, relu_cols=0, residual=None, res_upsample=False,
                in_scale=None, in_shift=None):
    """x [N,H,W,Cin], w [Cout,R,S,Cin] (packed layout) -> [N,OH,OW,Cout]."""
    # dtype follows the inputs: fp32 for the per-op parity tests, fp64 where a test wants an
    # "exact" value to measure fp32-grade error against
    if in_scale is not None:
        x = torch.relu(x * in_scale[:, None, None, :] + in_shift[:, None, None, :])
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), bias, stride=stride, padding=pad, dilation=dil)
    if residual is not None:
        r = residual.permute(0, 3, 1, 2)
        if res_upsample:
            r = F.interpolate(r, size=y.shape[-2:], mode="nearest")
        y = y + r
    y = y.permute(0, 2, 3, 1).contiguous()
    if relu_cols:
        y[..., :relu_cols] = torch.relu(y[..., :relu_cols])
    return y


def maxpool3x3s2_nhwc(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()


def groupnorm_affine(x, gamma, beta, groups=32, eps=1e-5):
    """Returns (scale, shift) [N,C] with group_norm(x) == x*scale + shift."""
    n, h, w, c = x.shape
    xg = x.double().reshape(n, h * w, groups, c // groups)
    mean = xg.mean(dim=(1, 3))
    var = xg.var(dim=(1, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    rstd_c = rstd.repeat_interleave(c // groups, dim=1)
    mean_c = mean.repeat_interleave(c // groups, dim=1)
    scale = gamma.double()[None] * rstd_c
    shift = beta.double()[None] - mean_c * scale
    return scale.float(), shift.float()


def groupnorm_relu_nhwc(x, gamma, beta, groups=32, eps=1e-5):
    y = F.group_norm(x.permute(0, 3, 1, 2), groups, gamma, beta, eps)
    return torch.relu(y).permute(0, 2, 3, 1).contiguous()

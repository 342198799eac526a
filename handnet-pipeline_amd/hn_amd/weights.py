"""Load-time weight transforms: BatchNorm folding and NCHW -> [Cout][R][S][Cin] repacking.

Done once on the host in fp64 (then rounded to fp32) when a reference-layout state_dict
is loaded; the device only ever sees folded, NHWC-ordered, channel-padded filters.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

BN_EPS = 1e-5


@dataclass
class ConvW:
    """A packed convolution: w [Cout,R,S,Cin_pad] fp32, bias [Cout] fp32 or None."""
    w: torch.Tensor
    bias: torch.Tensor | None
    stride: int = 1
    pad: int = 0
    dil: int = 1
    w16: torch.Tensor | None = None  # split-fp16 filter bank for hn_conv2d_nhwc_f16x3 (Cin % 32 == 0 only)

    def to(self, device):
        if self.w16 is not None:  # packed by hand (stem): keep
            w16 = self.w16.to(device)
        else:
            w16 = split_f16x3(self.w).to(device) if self.w.shape[3] % 32 == 0 else None
        return ConvW(self.w.to(device).contiguous(), None if self.bias is None else self.bias.to(device).contiguous(),
                     self.stride, self.pad, self.dil, w16)

    @property
    def cout(self):
        return self.w.shape[0]

    @property
    def cin(self):
        return self.w.shape[3]

    def macs_per_pixel(self, real_cin=None):
        r, s = self.w.shape[1], self.w.shape[2]
        return self.cout * r * s * (real_cin or self.cin)


F16_MAX = 65504.0


def check_split_range(w: torch.Tensor, what: str) -> None:
    """f16x3 range contract for weights: hi = fp16(w) must be finite (BatchNorm folding can blow a filter up by
    gamma / sqrt(var + eps)).  Checked once at load, on the host; such a model needs precision='f32'."""
    if w.numel() and not bool(torch.isfinite(w).all() and w.abs().max() <= F16_MAX):
        raise ValueError(f"{what}: values outside the fp16 range (|w| max {float(w.abs().max()):.3g} > {F16_MAX}) or "
                         "non-finite -- the f16x3 split format cannot hold them; build the engine with precision='f32'")


def split_f16x3(w: torch.Tensor) -> torch.Tensor:
    """[Cout,R,S,Cin] fp32 -> fp16 [Cout, (Cin/32)*R*S, 2, 32]: k tiles ordered 32-channel block
    OUTER / filter tap INNER (the order the f16x3 kernel walks K, chosen for L2 reuse of the input);
    per tile the hi run then the lo run, hi = fp16(w), lo = fp16(w - hi) (22 significant bits)."""
    cout, r, s, cin = w.shape
    if cin % 32:
        raise ValueError("split_f16x3 needs Cin to be a multiple of 32")
    check_split_range(w, "filter bank")
    tiles = w.float().reshape(cout, r * s, cin // 32, 32).permute(0, 2, 1, 3).reshape(cout, -1, 32)
    hi = tiles.half()
    lo = (tiles - hi.float()).half()
    return torch.stack([hi, lo], dim=2).contiguous()


def pack_stem_split(weight, bn=None, stride=2) -> "ConvW":
    """R x R stem conv ([Cout, Cin<=4, R, R], optional BN fold) for hn_conv_stem_f16x3: each filter ROW becomes
    one 32-deep k tile with k = kx*4 + c (zeros for k >= 4R), stored as fp16 [Cout][R][2][32] (hi run, lo run)."""
    w = weight.double()
    cout, cin, r, s = w.shape
    if r != s or r > 8 or cin > 4:
        raise ValueError("stem filter must be R x R with R <= 8 and Cin <= 4")
    bias = None
    if bn is not None:
        scale, shift = bn
        w = w * scale.view(-1, 1, 1, 1)
        bias = shift.float()
    rows = torch.zeros((cout, r, 1, 32), dtype=torch.float64)
    rows[:, :, 0, :4 * r].view(cout, r, r, 4)[..., :cin] = w.permute(0, 2, 3, 1)  # [o, ky, kx, c]
    nhwc = torch.zeros((cout, r, r, 4), dtype=torch.float32)
    nhwc[..., :cin] = w.permute(0, 2, 3, 1).float()
    return ConvW(nhwc.contiguous(), bias, stride, r // 2, 1, split_f16x3(rows.float()))


def _pad_to(c: int, m: int) -> int:
    return (c + m - 1) // m * m


def bn_scale_shift(sd, name, eps=BN_EPS):
    """(Frozen)BatchNorm2d eval: y = x*scale + shift (fp64)."""
    w = sd[name + ".weight"].double()
    b = sd[name + ".bias"].double()
    rm = sd[name + ".running_mean"].double()
    rv = sd[name + ".running_var"].double()
    scale = w / torch.sqrt(rv + eps)
    return scale, b - rm * scale


def pack_conv(weight, bias=None, bn=None, stride=1, pad=0, dil=1, cin_pad_to=4, sum_cin=False) -> ConvW:
    """weight [Cout,Cin,R,S] (torch layout) -> ConvW.

    bn        (scale, shift) from bn_scale_shift, folded into weight/bias
    sum_cin   collapse the input channels into one (the A2J stem sees the depth map
              replicated 3x -- a2j/a2j.py:199 -- so conv(w, [d,d,d]) == conv(sum_c w_c, d))
    """
    w = weight.double()
    b = bias.double() if bias is not None else None
    if sum_cin:
        w = w.sum(dim=1, keepdim=True)
    if bn is not None:
        scale, shift = bn
        w = w * scale[:, None, None, None]
        b = shift if b is None else b * scale + shift
    cout, cin, r, s = w.shape
    cp = _pad_to(cin, cin_pad_to)
    packed = torch.zeros((cout, r, s, cp), dtype=torch.float64)
    packed[..., :cin] = w.permute(0, 2, 3, 1)
    return ConvW(packed.float().contiguous(), None if b is None else b.float().contiguous(), stride, pad, dil)


def concat_cout(convs) -> ConvW:
    """Stack several convs that read the same input along Cout (same geometry)."""
    c0 = convs[0]
    for c in convs[1:]:
        assert c.w.shape[1:] == c0.w.shape[1:] and (c.stride, c.pad, c.dil) == (c0.stride, c0.pad, c0.dil)
    w = torch.cat([c.w for c in convs], 0)
    if all(c.bias is None for c in convs):
        b = None
    else:
        b = torch.cat([c.bias if c.bias is not None else torch.zeros(c.cout) for c in convs], 0)
    return ConvW(w.contiguous(), b, c0.stride, c0.pad, c0.dil)


def strip_prefix(sd, prefix):
    """Lightning checkpoints store A2J under 'a2j.' (a2j/a2j.py:277)."""
    if any(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    return sd

"""Torch-facing wrappers over the C ABI: allocate outputs with torch, launch on the
current HIP stream.  PyTorch is only the allocator / stream provider here.

Every function requires CUDA(HIP) tensors and raises if the extension is unavailable;
there is deliberately no CPU or eager fallback.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, ConvGroup, FcosLevels, check, ptr


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw handle of the current HIP stream.  torch.cuda.current_stream() builds a Stream object through four
    Python layers (40 % of the host time of an eager batch-1 step); the C accessors behind it are used directly."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _check_device(t: torch.Tensor, name="tensor"):
    """Launches go to the CURRENT device's stream: a tensor of another GPU would be a memory fault (or silently
    run on the wrong device under peer access).  The engines enter `on_device(self.device)`; direct callers of
    these wrappers must make the tensor's device current (torch.cuda.device / set_device)."""
    if _cur_device is not None and t.device.index != _cur_device():
        raise RuntimeError(f"{name} lives on cuda:{t.device.index} but the current device is cuda:{_cur_device()}; "
                           "wrap the call in `with torch.cuda.device(tensor.device):`")


class on_device:
    """`with on_device(dev):` -- make `dev` current for the block (no-op, and no torch.cuda.device object, when
    it already is: the hot path stays free of the context-manager cost)."""

    __slots__ = ("idx", "ctx")

    def __init__(self, dev):
        self.idx = torch.device(dev).index
        self.ctx = None

    def __enter__(self):
        if self.idx is not None and (_cur_device is None or _cur_device() != self.idx):
            self.ctx = torch.cuda.device(self.idx)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


def device_guarded(fn):
    """Method decorator: run with `self.device` as the current device."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with on_device(self.device):
            return fn(self, *a, **k)
    return wrapper


def _req(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (no CPU fallback in handnet-pipeline_amd)")
    _check_device(t, name)
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def conv_out_size(h, w, r, s, stride, pad, dil):
    return ((h + 2 * pad - dil * (r - 1) - 1) // stride + 1,
            (w + 2 * pad - dil * (s - 1) - 1) // stride + 1)


def is_split(t) -> bool:
    """S32 split activation tensor: fp16 [N,H,W,C/32,2,32] (hi[32] | lo[32] per 32-channel block)."""
    return t is not None and t.dtype == torch.float16 and t.dim() == 6 and t.shape[4] == 2 and t.shape[5] == 32


def channels(t) -> int:
    return t.shape[3] * 32 if is_split(t) else t.shape[3]


def _pixel_stride(t: torch.Tensor, name: str) -> int:
    """Accept a dense NHWC tensor (fp32 or S32) or a channel slice of one (x[..., c0:c1] /
    xs[:, :, :, b0:b1]).  Returns the pixel stride in elements (floats or halfs)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (no CPU fallback in handnet-pipeline_amd)")
    _check_device(t, name)
    n, h, w = t.shape[:3]
    ps = t.stride(2)
    if is_split(t):
        inner_ok = t.stride(5) == 1 and t.stride(4) == 32 and t.stride(3) == 64 and ps >= 64 * t.shape[3]
    elif t.dtype == torch.float32 and t.dim() == 4:
        inner_ok = t.stride(3) == 1 and ps >= t.shape[3]
    else:
        raise TypeError(f"{name}: expected fp32 [N,H,W,C] or fp16 S32 [N,H,W,C/32,2,32], got {t.dtype} {tuple(t.shape)}")
    if not inner_ok or t.stride(1) != w * ps or t.stride(0) != h * w * ps:
        raise ValueError(f"{name} must be NHWC-dense or a channel slice of an NHWC-dense tensor")
    return ps


_DESC_TEMPLATES = {}
_MULTI_PLANS = {}
_CONV_PLANS = {}


def make_conv_desc(n, h, w, cin, cout, r, s, stride=1, pad=0, dil=1, relu_cols=0, res_mode=0,
                   res_h=0, res_w=0, in_affine=0, tile=0, in_pix_stride=0, out_pix_stride=0) -> ConvDesc:
    """A fresh (mutable) descriptor.  The 25-keyword ctypes constructor costs ~5 us, more than the launch it
    describes at batch 1, so descriptors are stamped from cached templates."""
    key = (n, h, w, cin, cout, r, s, stride, pad, dil, relu_cols, res_mode, res_h, res_w, in_affine, tile,
           in_pix_stride, out_pix_stride)
    tpl = _DESC_TEMPLATES.get(key)
    if tpl is None:
        oh, ow = conv_out_size(h, w, r, s, stride, pad, dil)
        tpl = _DESC_TEMPLATES[key] = ConvDesc(
            n=n, h=h, w=w, cin=cin, cout=cout, r=r, s=s, stride=stride, pad=pad, dil=dil, oh=oh, ow=ow,
            relu_cols=relu_cols, res_mode=res_mode, res_h=res_h, res_w=res_w, in_affine=in_affine, tile=tile,
            out_split=0, res_split=0, res_pix_stride=0, in_pix_stride=in_pix_stride, out_pix_stride=out_pix_stride,
            in_affine_stride=0, splitk=0)
    return ConvDesc.from_buffer_copy(tpl)


def to_split(x, scale=None, shift=None, relu=False, out=None):
    """fp32 [N,H,W,C] -> S32 [N,H,W,C/32,2,32]; optional per-(image, channel) affine (+ReLU) first
    (GroupNorm-apply of the FCOS towers)."""
    lib = _lib.load()
    xs = _pixel_stride(x, "x")
    if is_split(x):
        raise TypeError("x is already split")
    n, h, w, c = x.shape
    if c % 32:
        raise ValueError("the S32 format needs C % 32 == 0")
    if out is None:
        out = torch.empty((n, h, w, c // 32, 2, 32), device=x.device, dtype=torch.float16)
    ys = _pixel_stride(out, "out")
    a_stride = 0
    if scale is not None:
        for t in (scale, shift):
            if (not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (n, c) or t.stride(1) != 1
                    or t.stride(0) != scale.stride(0)):
                raise ValueError("scale / shift must be fp32 GPU [N, C] tables with equal row stride")
        a_stride = scale.stride(0)
    check(lib.hn_affine_split_f32(ptr(x), ptr(scale), ptr(shift), 1 if relu else 0, n, h * w, c, xs, a_stride,
                                  ptr(out), ys, _stream()), "hn_affine_split_f32")
    return out


def from_split(xs16, out=None):
    """S32 -> fp32 [N,H,W,C] (exact)."""
    lib = _lib.load()
    ps = _pixel_stride(xs16, "x")
    if not is_split(xs16):
        raise TypeError("x is not an S32 tensor")
    n, h, w, nb = xs16.shape[:4]
    if out is None:
        out = torch.empty((n, h, w, nb * 32), device=xs16.device, dtype=torch.float32)
    check(lib.hn_unsplit_f32(ptr(xs16), n, h * w, nb * 32, ps, ptr(out), _pixel_stride(out, "out"), _stream()),
          "hn_unsplit_f32")
    return out


# bench.py's roofline leg: when set to a list, every conv launch is bracketed by HIP events
# on the launch stream and ((precision, tile id), algorithmic MACs, timer, shape) is appended.
CONV_PROFILE = None
# which stage of the hot path the engines are in (bench.py's roofline.stages): "resnet34_body", "fpn", "towers",
# "head_outputs", "a2j_trunk", "a2j_heads"; recorded with every profiled conv launch
PROFILE_STAGE = None
TILE_NAMES = {1: "128x128", 2: "128x64", 3: "64x64", 4: "128x32", 6: "64x128", 7: "32x64", 8: "256x64", 12: "64x64k2"}   # (5, 9-11: retired sweep forms)
TILE_RS = 0x100  # profile records: tile id | TILE_RS when the launch ran the row-shared-A instantiation of that tile
TILE_HALO = 0x200  # ... | TILE_HALO when it was routed to the halo-patch kernel (conv3x3_halo_kernel)
TILE_MULTI = 0x400  # ... | TILE_MULTI for a heterogeneous launch (conv_igemm_f16x3_multi_kernel): several convolutions, one record
TILE_STREAM = 0x800  # ... | TILE_STREAM when it was routed to the streaming 1x1 kernel (conv1x1_stream_kernel)


def tile_name(tile) -> str:
    """'128x128' / '128x128+rs' (the row-shared-A kernels are separate instantiations, i.e. separate profiler rows)."""
    if isinstance(tile, int) and tile & TILE_HALO:
        return "halo16x16"
    if isinstance(tile, int) and tile & TILE_STREAM:
        return "stream1x1"
    base = TILE_NAMES.get(tile & 0xFF, str(tile & 0xFF)) if isinstance(tile, int) else str(tile)
    return base + ("+rs" if isinstance(tile, int) and tile & TILE_RS else "") + \
        ("+multi" if isinstance(tile, int) and tile & TILE_MULTI else "")


# Term count of the f16x3 kernels for the launches issued from now on: 3 = the split-precision product (default),
# 1 = hi*hi only (the engines' precision="f16x1" throughput mode, reported beside the headline by bench.py).
F16_TERMS = 3


class f16_terms:
    """`with ops.f16_terms(1):` -- launches inside the block carry hn_conv_desc.terms = 1."""

    def __init__(self, terms):
        self.terms = terms

    def __enter__(self):
        global F16_TERMS
        self._old, F16_TERMS = F16_TERMS, self.terms

    def __exit__(self, *a):
        global F16_TERMS
        F16_TERMS = self._old


def clear_plan_caches():
    """Drop the cached conv descriptors / launch plans (they are keyed by weight addresses: engines call this when
    they are rebuilt, so that a recycled address can never meet a stale plan and the caches stay bounded)."""
    _CONV_PLANS.clear()
    _DESC_TEMPLATES.clear()
    _MULTI_PLANS.clear()


# split-K workspace: one fp32 buffer per (device, stream) -- a convolution only uses it between its own two
# launches, and launches on one stream are ordered
CONV_WORKSPACE_BYTES = 32 << 20
SPLITK = True  # development switch (forms.apply_env: HN_SPLITK=0)
# split short k loops too (desc.splitk = 1).  It used to pay only under graph replay; since the host path got
# cheaper (raw stream handle, cached descriptors) it also wins in eager mode (batch 1: 294 -> 301 frames/s)
SPLITK_EAGER = True


class launch_cost_hidden:
    """Context for code whose launches will be replayed from a hipGraph: forces the aggressive split-K setting
    (the default since the host path became cheap; HN_SPLITK_EAGER=0 restores the conservative eager rule)."""

    def __enter__(self):
        global SPLITK_EAGER
        self._old, SPLITK_EAGER = SPLITK_EAGER, True

    def __exit__(self, *a):
        global SPLITK_EAGER
        SPLITK_EAGER = self._old
_WORKSPACES = {}


def _conv_workspace(device):
    key = (device.index, _stream())
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = torch.empty((CONV_WORKSPACE_BYTES // 4,), device=device, dtype=torch.float32)
    return ws


def conv2d_nhwc(x, w, bias=None, *, stride=1, pad=0, dil=1, relu=False, relu_cols=None,
                residual=None, res_upsample=False, in_scale=None, in_shift=None, out=None, tile=0,
                algo_cin=None, w16=None, out_split=False, gn_partial=None, splitk=True, force_splits=None):
    """Convolution with fused epilogue.  x: fp32 [N,H,W,Cin] or S32 split; w [Cout,R,S,Cin] fp32.

    w16 given  -> f16x3 kernel (split-fp16 operands on the f16 MFMA, fp32-grade results); an fp32
                  x (and a GroupNorm-on-load affine in_scale/in_shift) is first converted with
                  one hn_affine_split_f32 pass.  Otherwise the exact f32-MFMA kernel runs.
    out_split  -> y is written in the S32 format (Cout % 32 == 0), ready for the next f16x3 conv.
    residual   -> fp32 or S32 tensor added before the ReLU; res_upsample = nearest-neighbour
                  read of a coarser map (FPN top-down path).
    algo_cin   -> input channels the reference's conv really has when Cin is zero-padded
                  (FLOP accounting only).
    splitk     -> (f16x3) allow split-K for small grids (deterministic two-launch scheme, needs the workspace).
    gn_partial -> (f16x3, fp32 output, no residual / ReLU) fp32 scratch of gn_rows32_scratch_floats(rows, Cout)
                  floats: the epilogue also writes GroupNorm partial sums for groupnorm_finalize_rows32()."""
    lib = _lib.load()
    # Plan cache: a call whose tensors have the shapes / strides / flags of an earlier call reuses that call's
    # validated descriptor (the checks below cost more host time than the launch at batch 1).
    plan_key = None
    if (out is None and in_scale is None and w16 is not None and gn_partial is None and CONV_PROFILE is None
            and x.dtype == torch.float16 and x.is_cuda):
        _check_device(x, "x")
        plan_key = (x.device.index, x.shape, x.stride(), w.data_ptr(), tuple(w.shape), w16.data_ptr(),
                    None if bias is None else bias.data_ptr(),
                    stride, pad, dil, relu, relu_cols, tile, out_split, res_upsample, splitk, SPLITK, SPLITK_EAGER, F16_TERMS,
                    None if residual is None else (residual.shape, residual.stride(), residual.dtype))
        plan = _CONV_PLANS.get(plan_key)
        if plan is not None:
            tpl, out_shape, out_dtype, use_ws = plan
            d = ConvDesc.from_buffer_copy(tpl)
            out = torch.empty(out_shape, device=x.device, dtype=out_dtype)
            ws = _conv_workspace(x.device) if use_ws else None
            check(lib.hn_conv2d_nhwc_f16x3_ws(C.byref(d), ptr(x), ptr(w16), ptr(bias), ptr(residual), ptr(out),
                                              ptr(ws), ws.numel() * 4 if use_ws else 0, _stream()),
                  "hn_conv2d_nhwc_f16x3_ws")
            return out
    _req(w, name="w")
    cout, r, s, cin = w.shape
    use16 = w16 is not None
    if use16:
        if is_split(x):
            if in_scale is not None:
                raise ValueError("apply the affine with to_split() before an S32-input convolution")
        else:
            x = to_split(x, in_scale, in_shift, relu=in_scale is not None)
            in_scale = in_shift = None
    elif is_split(x):
        raise TypeError("the f32 kernel needs an fp32 input (use from_split)")
    xs = _pixel_stride(x, "x")
    n, h, wd = x.shape[:3]
    if channels(x) != cin:
        raise ValueError(f"weight Cin {cin} != input channels {channels(x)}")
    rc = cout if relu else 0
    if relu_cols is not None:
        rc = relu_cols
    res_mode, rh, rw = 0, 0, 0
    if residual is not None:
        if channels(residual) != cout or residual.shape[0] != n:
            raise ValueError("residual shape mismatch")
        if is_split(residual) and not use16:
            raise TypeError("the f32 kernel takes fp32 residuals only")
        if res_upsample:
            res_mode, rh, rw = 2, residual.shape[1], residual.shape[2]
        else:
            res_mode = 1
    dense_in = xs == (2 * cin if is_split(x) else cin)
    d = make_conv_desc(n, h, wd, cin, cout, r, s, stride, pad, dil, rc, res_mode, rh, rw,
                       1 if in_scale is not None else 0, tile, in_pix_stride=0 if dense_in else xs)
    if out is None:
        if out_split:
            if cout % 32:
                raise ValueError("out_split needs Cout % 32 == 0")
            out = torch.empty((n, d.oh, d.ow, cout // 32, 2, 32), device=x.device, dtype=torch.float16)
        else:
            out = torch.empty((n, d.oh, d.ow, cout), device=x.device, dtype=torch.float32)
    if tuple(out.shape[:3]) != (n, d.oh, d.ow) or channels(out) != cout:
        raise ValueError("out has the wrong shape")
    ys = _pixel_stride(out, "out")
    d.out_split = 1 if is_split(out) else 0
    d.out_pix_stride = 0 if ys == (2 * cout if d.out_split else cout) else ys
    d.terms = 1 if (use16 and F16_TERMS == 1) else 0
    if residual is not None:
        rs = _pixel_stride(residual, "residual")
        d.res_split = 1 if is_split(residual) else 0
        if res_mode == 1 and tuple(residual.shape[:3]) != (n, d.oh, d.ow):
            raise ValueError("residual shape mismatch")
        if use16:
            d.res_pix_stride = rs
        elif rs != cout:
            raise ValueError("the f32 kernel needs a dense residual")
    if bias is not None:
        _req(bias, name="bias")
    if in_scale is not None:
        # [N, Cin] tables, possibly column slices of a wider [N, C] table
        for t in (in_scale, in_shift):
            if (not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (n, cin) or t.stride(1) != 1
                    or t.stride(0) != in_scale.stride(0)):
                raise ValueError("in_scale / in_shift must be fp32 GPU [N, Cin] tables with equal row stride")
        d.in_affine_stride = 0 if in_scale.stride(0) == cin else in_scale.stride(0)
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    if use16:
        if (not w16.is_cuda or w16.dtype != torch.float16 or not w16.is_contiguous()
                or w16.numel() != 2 * w.numel()):
            raise ValueError("w16 must be the contiguous fp16 GPU tensor produced by weights.split_f16x3(w)")
        if gn_partial is not None:
            if residual is not None or rc or d.out_split:
                raise ValueError("gn_partial needs an fp32 output without residual / ReLU")
            need = lib.hn_groupnorm_rows32_scratch_floats(n * d.oh * d.ow, cout)
            if not gn_partial.is_cuda or gn_partial.dtype != torch.float32 or gn_partial.numel() < need:
                raise ValueError(f"gn_partial must be an fp32 GPU buffer of >= {need} floats")
            check(lib.hn_conv2d_nhwc_f16x3_gn(C.byref(d), ptr(x), ptr(w16), ptr(bias), ptr(out), ptr(gn_partial),
                                              _stream()), "hn_conv2d_nhwc_f16x3_gn")
        else:
            splitk = splitk and SPLITK
            d.splitk = (1 if SPLITK_EAGER else 0) if splitk else -1
            if force_splits is not None:      # sweeps only: exactly this many splits (1 = none)
                d.splitk = int(force_splits) if force_splits >= 2 else -1
                plan_key = None
            ws = _conv_workspace(x.device) if splitk else None
            if plan_key is not None:
                if bias is not None and not bias.is_contiguous():
                    plan_key = None
                else:
                    _CONV_PLANS[plan_key] = (ConvDesc.from_buffer_copy(d), tuple(out.shape), out.dtype, bool(splitk))
            check(lib.hn_conv2d_nhwc_f16x3_ws(C.byref(d), ptr(x), ptr(w16), ptr(bias), ptr(residual), ptr(out),
                                              ptr(ws), ws.numel() * 4 if splitk else 0, _stream()),
                  "hn_conv2d_nhwc_f16x3_ws")
    else:
        if gn_partial is not None:
            raise ValueError("gn_partial is an f16x3-kernel feature")
        check(lib.hn_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(in_scale),
                                     ptr(in_shift), ptr(out), _stream()), "hn_conv2d_nhwc_f32")
    if prof is not None:
        timer.stop()
        macs = n * d.oh * d.ow * cout * r * s * (algo_cin or cin)
        if use16:
            flags = TILE_HALO if lib.hn_conv2d_f16x3_uses_halo(C.byref(d), 1 if residual is not None else 0) else \
                (TILE_STREAM if lib.hn_conv2d_f16x3_uses_stream(C.byref(d)) else
                 TILE_RS if lib.hn_conv2d_f16x3_uses_rs(C.byref(d)) else 0)
            kind = ("f16x3", lib.hn_conv2d_f16x3_pick_tile(C.byref(d)) | flags)
        else:
            kind = ("f32", lib.hn_conv2d_pick_tile(C.byref(d)))
        prof.append((kind, macs, timer, (n, h, wd, cin, cout, r, stride, dil), PROFILE_STAGE))
    return out


def conv2d_nhwc_multi(items, fused=None):
    """INDEPENDENT f16x3 convolutions of different shapes as one launch (hn_conv2d_nhwc_f16x3_multi; at most
    _lib.CONV_MULTI_MAX members).  items: [(x S32, ConvW-like cw, opts)], opts = dict(relu=False, relu_cols=None,
    residual=None, out_split=True); every member computes exactly what conv2d_nhwc(x, cw.w, cw.bias, stride=cw.stride,
    pad=cw.pad, dil=cw.dil, w16=cw.w16, **opts) would (bit-identical: same kernel body, same split-K plan) -- the library
    falls back to that, member after member, when the members' tile forms differ.  Returns the outputs.
    fused: optional one-element list that receives whether the members ran as one launch (tests)."""
    lib = _lib.load()
    k = len(items)
    if k == 0 or k > _lib.CONV_MULTI_MAX:
        raise ValueError(f"need 1..{_lib.CONV_MULTI_MAX} members")
    splitk = SPLITK
    # Plan cache (like conv2d_nhwc's): the validated descriptor table of an earlier call with the same shapes / strides /
    # weights is stamped out again; only the activation pointers change.
    key = [splitk, SPLITK_EAGER, F16_TERMS]
    for x, cw, opts in items:
        res = opts.get("residual")
        key.append((x.device.index, x.shape, x.stride(), x.dtype, cw.w16.data_ptr() if cw.w16 is not None else None,
                    None if cw.bias is None else cw.bias.data_ptr(), cw.stride, cw.pad, cw.dil, opts.get("relu", False),
                    opts.get("relu_cols"), opts.get("out_split", True),
                    None if res is None else (res.shape, res.stride(), res.dtype)))
    key = tuple(key)
    plan = _MULTI_PLANS.get(key) if CONV_PROFILE is None and fused is None else None
    if plan is not None:
        tpl, out_specs, use_ws = plan
        mm = _lib.ConvMulti.from_buffer_copy(tpl)
        outs = []
        for i, ((x, _cw, opts), (shape, dtype)) in enumerate(zip(items, out_specs)):
            _check_device(x, "x")
            y = torch.empty(shape, device=x.device, dtype=dtype)
            mm.x16[i], mm.residual[i], mm.y[i] = x.data_ptr(), ptr(opts.get("residual")), y.data_ptr()
            outs.append(y)
        ws = _conv_workspace(items[0][0].device) if use_ws else None
        check(lib.hn_conv2d_nhwc_f16x3_multi(C.byref(mm), ptr(ws), ws.numel() * 4 if use_ws else 0, _stream()),
              "hn_conv2d_nhwc_f16x3_multi")
        return outs
    mm = _lib.ConvMulti()
    mm.count = k
    outs = []
    macs, shapes = 0, []
    for i, (x, cw, opts) in enumerate(items):
        if not is_split(x) or cw.w16 is None:
            raise TypeError("multi launches take S32 inputs and split filter banks")
        if not x.is_cuda:
            raise RuntimeError("x must live on the GPU (no CPU fallback in handnet-pipeline_amd)")
        _check_device(x, "x")
        relu, relu_cols = opts.get("relu", False), opts.get("relu_cols")
        residual, out_split = opts.get("residual"), opts.get("out_split", True)
        cout, r, s_, cin = cw.w.shape
        n, h, wd = x.shape[:3]
        if channels(x) != cin:
            raise ValueError(f"member {i}: weight Cin {cin} != input channels {channels(x)}")
        xs = _pixel_stride(x, "x")
        rc = (cout if relu else 0) if relu_cols is None else relu_cols
        d = make_conv_desc(n, h, wd, cin, cout, r, s_, cw.stride, cw.pad, cw.dil, rc, 1 if residual is not None else 0, 0, 0,
                           0, 0, in_pix_stride=0 if xs == 2 * cin else xs)
        if out_split:
            if cout % 32:
                raise ValueError("out_split needs Cout % 32 == 0")
            y = torch.empty((n, d.oh, d.ow, cout // 32, 2, 32), device=x.device, dtype=torch.float16)
        else:
            y = torch.empty((n, d.oh, d.ow, cout), device=x.device, dtype=torch.float32)
        d.out_split = 1 if out_split else 0
        d.terms = 1 if F16_TERMS == 1 else 0
        d.splitk = (1 if SPLITK_EAGER else 0) if splitk else -1
        if residual is not None:
            if tuple(residual.shape[:3]) != (n, d.oh, d.ow) or channels(residual) != cout:
                raise ValueError(f"member {i}: residual shape mismatch")
            d.res_split = 1 if is_split(residual) else 0
            d.res_pix_stride = _pixel_stride(residual, "residual")
        if cw.bias is not None:
            _req(cw.bias, name="bias")
        mm.desc[i] = d
        mm.x16[i], mm.w16[i], mm.bias[i] = ptr(x), ptr(cw.w16), ptr(cw.bias)
        mm.residual[i], mm.y[i] = ptr(residual), ptr(y)
        outs.append(y)
        macs += n * d.oh * d.ow * cout * r * s_ * cin
        shapes.append((n, h, wd, cin, cout, r, cw.stride, cw.dil))
    ws = _conv_workspace(items[0][0].device) if splitk else None
    if fused is not None:
        fused.append(bool(lib.hn_conv2d_f16x3_multi_fuses(C.byref(mm), ws.numel() * 4 if splitk else 0)))
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv2d_nhwc_f16x3_multi(C.byref(mm), ptr(ws), ws.numel() * 4 if splitk else 0, _stream()),
          "hn_conv2d_nhwc_f16x3_multi")
    if prof is not None:
        timer.stop()
        big = max(range(k), key=lambda i: shapes[i][0] * shapes[i][1] * shapes[i][2] * shapes[i][3] * shapes[i][4] * shapes[i][5] ** 2)
        prof.append((("f16x3", lib.hn_conv2d_f16x3_pick_tile(C.byref(mm.desc[big])) | TILE_MULTI), macs, timer, shapes[big],
                     PROFILE_STAGE))
    else:
        _MULTI_PLANS[key] = (_lib.ConvMulti.from_buffer_copy(mm), [(tuple(y.shape), y.dtype) for y in outs], bool(splitk))
    return outs


def conv2d_nhwc_grouped(xs, ws, *, pad=0, relu=False, relu_cols=None, out_split=False, tile=0, gn_partials=None,
                        outs=None, out_channel_offsets=None, gn=None, gn_units=0):
    """Independent stride-1 f16x3 convolutions with identical channels / filter / batch as ONE launch
    (gridDim.z = member); the members may differ in spatial size (the FPN levels of one layer).
      xs            S32 inputs (same N, channels and pixel stride); ws: ConvW-like objects (.w [Cout,R,S,Cin], .bias,
                    .w16), all of one shape
      outs          optional preallocated outputs (one per member, or the same tensor several times); member i writes
                    channels [out_channel_offsets[i], +Cout) of outs[i] -- stacking members in one wide tensor keeps
                    e.g. the cls / reg towers of a level together.  Default: fresh dense outputs.
      gn_partials   one GroupNorm-sum buffer per member (each Cout/8 units wide), or
      gn, gn_units  [(buffer, unit offset)] per member into slabs that are gn_units units wide (stacked members).
    Returns the list of outputs.  A single plain member falls through to conv2d_nhwc."""
    k = len(xs)
    if k != len(ws) or k == 0 or k > _lib.CONV_MAX_GROUP:
        raise ValueError(f"need 1..{_lib.CONV_MAX_GROUP} inputs and as many weight sets")
    if k == 1 and outs is None and gn is None:
        return [conv2d_nhwc(xs[0], ws[0].w, ws[0].bias, pad=pad, relu=relu, relu_cols=relu_cols, w16=ws[0].w16,
                            out_split=out_split, tile=tile, gn_partial=None if gn_partials is None else gn_partials[0])]
    lib = _lib.load()
    cout, r, s, cin = ws[0].w.shape
    x0 = xs[0]
    xstride = _pixel_stride(x0, "x")
    n = x0.shape[0]
    for x, cw in zip(xs, ws):
        if not is_split(x) or x.shape[0] != n or channels(x) != cin or _pixel_stride(x, "x") != xstride:
            raise ValueError("grouped inputs must be S32 tensors of one batch size, channel count and pixel stride")
        if tuple(cw.w.shape) != (cout, r, s, cin) or cw.w16 is None or (cw.bias is None) != (ws[0].bias is None):
            raise ValueError("grouped weights must share one shape (and all or none have a bias)")
    rc = (cout if relu else 0) if relu_cols is None else relu_cols
    h0, w0 = x0.shape[1:3]
    d = make_conv_desc(n, h0, w0, cin, cout, r, s, 1, pad, 1, rc, 0, 0, 0, 0, tile,
                       in_pix_stride=0 if xstride == 2 * cin else xstride)
    d.splitk = -1
    d.terms = 1 if F16_TERMS == 1 else 0
    if gn_partials is not None:
        if gn is not None or len(gn_partials) != k:
            raise ValueError("give gn_partials (one per member) or gn, not both")
        gn, gn_units = [(g, 0) for g in gn_partials], 0
    if gn is not None and (rc or out_split):
        raise ValueError("GroupNorm sums need fp32 outputs without ReLU")
    sizes = [conv_out_size(x.shape[1], x.shape[2], r, s, 1, pad, 1) for x in xs]
    if outs is None:
        if out_split:
            outs = [torch.empty((n, oh, ow, cout // 32, 2, 32), device=x0.device, dtype=torch.float16) for oh, ow in sizes]
        else:
            outs = [torch.empty((n, oh, ow, cout), device=x0.device, dtype=torch.float32) for oh, ow in sizes]
    offs = [0] * k if out_channel_offsets is None else list(out_channel_offsets)
    ystride = _pixel_stride(outs[0], "out")
    d.out_split = 1 if is_split(outs[0]) else 0
    d.out_pix_stride = 0 if ystride == (2 * cout if d.out_split else cout) else ystride
    grp = ConvGroup()
    grp.count = k
    grp.gn_units = int(gn_units)
    units = gn_units if gn_units else cout // 8
    for i, (x, cw, y, (oh, ow)) in enumerate(zip(xs, ws, outs, sizes)):
        if (tuple(y.shape[:3]) != (n, oh, ow) or is_split(y) != bool(d.out_split) or _pixel_stride(y, "out") != ystride
                or offs[i] % 32 or offs[i] + cout > channels(y)):
            raise ValueError("grouped outputs must match their member's size and share type / pixel stride")
        elt = 2 if d.out_split else 4
        grp.x16[i], grp.w16[i], grp.bias[i] = ptr(x), ptr(cw.w16), ptr(cw.bias)
        grp.y[i] = y.data_ptr() + elt * (2 * offs[i] if d.out_split else offs[i])
        grp.h[i], grp.w[i] = x.shape[1], x.shape[2]
        grp.gn_partial[i] = None
        if gn is not None:
            buf, uoff = gn[i]
            need = lib.hn_groupnorm_rows32_scratch_floats(n * oh * ow, units * 8)
            if buf.numel() < need or buf.dtype != torch.float32 or not buf.is_cuda or uoff * 8 + cout > units * 8:
                raise ValueError(f"member {i}: GroupNorm slab must be an fp32 GPU buffer of >= {need} floats")
            grp.gn_partial[i] = buf.data_ptr() + 16 * uoff
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv2d_nhwc_f16x3_grouped(C.byref(d), C.byref(grp), _stream()), "hn_conv2d_nhwc_f16x3_grouped")
    if prof is not None:
        timer.stop()
        rows = sum(n * oh * ow for oh, ow in sizes)
        tot = make_conv_desc(1, rows, 1, cin, cout, 1, 1)          # what the tile heuristic saw: all rows together
        tile = lib.hn_conv2d_f16x3_pick_tile(C.byref(tot))
        q = ConvDesc.from_buffer_copy(d)                           # geometry of the launch, picked tile, narrowest member
        q.tile, q.w = tile, min(ow for _, ow in sizes)
        prof.append((("f16x3", tile | (TILE_RS if lib.hn_conv2d_f16x3_uses_rs(C.byref(q)) else 0)), rows * cout * r * s * cin,
                     timer, (1, rows, 1, cin, cout, r, 1, 1), PROFILE_STAGE))
    return outs


def conv3x3_thin_levels(xs, cw, relu_cols=0):
    """3x3 / pad 1 convolution with <= 16 output channels on several maps at once (the FCOS head outputs on all FPN levels,
    hn_conv3x3_thin_f16x3_levels): xs = S32 inputs [N,h_l,w_l,Cin/32,2,32] (channel slices allowed), cw = ConvW-like with
    .w [Cout,3,3,Cin], .w16, .bias -> list of fp32 [N,h_l,w_l,Cout].  Cout <= 5: the P-form kernel (equal to
    conv2d_nhwc_grouped(xs, [cw]*L, pad=1) to fp32 rounding); else the tap kernel (bit-identical to it)."""
    lib = _lib.load()
    cout, r, s, cin = cw.w.shape
    if (r, s) != (3, 3) or cout > 16 or cw.w16 is None or len(xs) > _lib.HN_FCOS_MAX_LEVELS:
        raise ValueError("conv3x3_thin_levels needs a 3x3 filter bank with <= 16 output channels and a split bank")
    n = xs[0].shape[0]
    xstride = _pixel_stride(xs[0], "x")
    lv = _lib.ThinLevels()
    lv.count = len(xs)
    outs = []
    for i, x in enumerate(xs):
        if not is_split(x) or x.shape[0] != n or channels(x) != cin or _pixel_stride(x, "x") != xstride:
            raise ValueError("levels must be S32 tensors of one batch size, channel count and pixel stride")
        y = torch.empty((n, x.shape[1], x.shape[2], cout), device=x.device, dtype=torch.float32)
        lv.x16[i], lv.y[i], lv.h[i], lv.w[i] = x.data_ptr(), y.data_ptr(), x.shape[1], x.shape[2]
        outs.append(y)
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv3x3_thin_f16x3_levels(C.byref(lv), n, cin, cout, ptr(cw.w16), ptr(cw.bias), int(relu_cols),
                                           0 if xstride == 2 * cin else xstride, _stream()), "hn_conv3x3_thin_f16x3_levels")
    if prof is not None:
        timer.stop()
        rows = sum(n * x.shape[1] * x.shape[2] for x in xs)
        form = "thin-P" if lib.hn_conv3x3_thin_uses_flat(C.byref(lv), n, cin, cout) else "thin16x16"
        prof.append((("f16x3", form), rows * cout * 9 * cin, timer, (1, rows, 1, cin, cout, 3, 1, 1), PROFILE_STAGE))
    return outs


def conv3x3_thin_levels_group(members):
    """Up to three conv3x3_thin_levels calls as ONE launch where they run the tap kernel (hn_conv3x3_thin_f16x3_levels_group; the
    FCOS head outputs of a single frame: 89 workgroups each, latency-bound): members = [(xs, cw, relu_cols), ...] with one batch
    size, channel count and pixel stride -> [outs of member 0, outs of member 1, ...], bit-identical to the separate calls (which
    is what runs where a member takes the P form)."""
    lib = _lib.load()
    if not 1 <= len(members) <= 3:
        raise ValueError("1..3 members")
    n = members[0][0][0].shape[0]
    cin = members[0][1].w.shape[3]
    xstride = _pixel_stride(members[0][0][0], "x")
    arr = (_lib.ThinMember * len(members))()
    results = []
    rows = 0
    for m, (xs, cw, relu_cols) in enumerate(members):
        cout, r, s_, c = cw.w.shape
        if (r, s_) != (3, 3) or cout > 16 or cw.w16 is None or len(xs) > _lib.HN_FCOS_MAX_LEVELS or c != cin:
            raise ValueError("members need 3x3 filter banks with <= 16 output channels, a split bank and one input channel count")
        mm = arr[m]
        mm.lv.count = len(xs)
        outs = []
        for i, x in enumerate(xs):
            if not is_split(x) or x.shape[0] != n or channels(x) != cin or _pixel_stride(x, "x") != xstride:
                raise ValueError("levels must be S32 tensors of one batch size, channel count and pixel stride")
            y = torch.empty((n, x.shape[1], x.shape[2], cout), device=x.device, dtype=torch.float32)
            mm.lv.x16[i], mm.lv.y[i], mm.lv.h[i], mm.lv.w[i] = x.data_ptr(), y.data_ptr(), x.shape[1], x.shape[2]
            outs.append(y)
        mm.cout, mm.relu_cols, mm.w16, mm.bias = cout, int(relu_cols), ptr(cw.w16), ptr(cw.bias)
        results.append(outs)
        rows += sum(n * x.shape[1] * x.shape[2] for x in xs) * cout
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv3x3_thin_f16x3_levels_group(arr, len(members), n, cin, 0 if xstride == 2 * cin else xstride, _stream()),
          "hn_conv3x3_thin_f16x3_levels_group")
    if prof is not None:
        timer.stop()
        flat = any(lib.hn_conv3x3_thin_uses_flat(C.byref(arr[m].lv), n, cin, arr[m].cout) for m in range(len(members)))
        px = sum(n * x.shape[1] * x.shape[2] for x in members[0][0])
        couts = sum(mm[1].w.shape[0] for mm in members)
        prof.append((("f16x3", "thin-P" if flat else f"thin16x16x{len(members)}"), rows * 9 * cin, timer,
                     (1, px, 1, cin, couts, 3, 1, 1), PROFILE_STAGE))
    return results


def _thin_levels_of(shapes):
    lv = _lib.ThinLevels()
    lv.count = len(shapes)
    for i, (h, w) in enumerate(shapes):
        lv.h[i], lv.w[i] = h, w
    return lv


def thin_affine_applies(ts, cw) -> bool:
    """Would conv3x3_thin_affine_levels take these raw tower outputs ts (fp32 [N,h_l,w_l,C]) with filter bank cw?"""
    cout, _, _, cin = cw.w.shape
    lv = _thin_levels_of([t.shape[1:3] for t in ts])
    return len(ts) <= _lib.HN_FCOS_MAX_LEVELS and bool(_lib.load().hn_conv3x3_thin_affine_applies(C.byref(lv), ts[0].shape[0], cin, cout))


def conv3x3_thin_affine_levels(ts, affines, ch0, cw, relu_cols=0):
    """The head-output convolution on the RAW outputs of the last tower layer with that layer's GroupNorm apply pass fused
    in (hn_conv3x3_thin_affine_f16x3_levels): ts[l] fp32 [N,h_l,w_l,C] raw conv outputs, affines[l] = (scale, shift) fp32
    [N,C] from groupnorm_finalize_rows32_levels, ch0 = first channel of this head's cw.cin channels inside C.  Equals
    conv3x3_thin_levels(to_split_levels(ts, affines)[..., ch0 // 32 : (ch0 + cin) // 32], cw) bit for bit, without the pass."""
    lib = _lib.load()
    cout, r, s, cin = cw.w.shape
    n, c = ts[0].shape[0], ts[0].shape[3]
    if (r, s) != (3, 3) or cw.w16 is None or ch0 % 32 or ch0 + cin > c:
        raise ValueError("conv3x3_thin_affine_levels needs a split 3x3 filter bank and a 32-aligned channel range")
    xstride = _pixel_stride(ts[0], "x")
    a_stride = affines[0][0].stride(0)
    lv, aff = _thin_levels_of([t.shape[1:3] for t in ts]), _lib.ThinAffine()
    aff.in_pix_stride, aff.affine_stride = xstride, a_stride
    outs = []
    for i, (t, (scale, shift)) in enumerate(zip(ts, affines)):
        if t.dtype != torch.float32 or is_split(t) or t.shape[0] != n or t.shape[3] != c or _pixel_stride(t, "x") != xstride:
            raise ValueError("levels must be fp32 NHWC tensors of one batch size, channel count and pixel stride")
        for q in (scale, shift):
            if not q.is_cuda or q.dtype != torch.float32 or tuple(q.shape) != (n, c) or q.stride(1) != 1 or q.stride(0) != a_stride:
                raise ValueError("scale / shift must be fp32 GPU [N, C] tables with equal row stride")
        y = torch.empty((n, t.shape[1], t.shape[2], cout), device=t.device, dtype=torch.float32)
        aff.x[i], aff.scale[i], aff.shift[i] = t.data_ptr() + 4 * ch0, scale.data_ptr() + 4 * ch0, shift.data_ptr() + 4 * ch0
        lv.y[i] = y.data_ptr()
        outs.append(y)
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv3x3_thin_affine_f16x3_levels(C.byref(lv), C.byref(aff), n, cin, cout, ptr(cw.w16), ptr(cw.bias),
                                                  int(relu_cols), _stream()), "hn_conv3x3_thin_affine_f16x3_levels")
    if prof is not None:
        timer.stop()
        rows = sum(n * t.shape[1] * t.shape[2] for t in ts)
        prof.append((("f16x3", "thin-P+gn"), rows * cout * 9 * cin, timer, (1, rows, 1, cin, cout, 3, 1, 1), PROFILE_STAGE))
    return outs


def thin_uses_flat(xs, cw) -> bool:
    """Would conv3x3_thin_levels(xs, cw) run the P-form kernel (True) or the tap kernel (False)?"""
    lv = _lib.ThinLevels()
    lv.count = len(xs)
    for i, x in enumerate(xs):
        lv.h[i], lv.w[i] = x.shape[1], x.shape[2]
    cout, _, _, cin = cw.w.shape
    return bool(_lib.load().hn_conv3x3_thin_uses_flat(C.byref(lv), xs[0].shape[0], cin, cout))


def maxpool3x3s2_nhwc(x, out=None):
    lib = _lib.load()
    if is_split(x):
        if not x.is_contiguous():
            raise ValueError("x must be contiguous")
        n, h, w, nb = x.shape[:4]
        oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        if out is None:
            out = torch.empty((n, oh, ow, nb, 2, 32), device=x.device, dtype=torch.float16)
        check(lib.hn_maxpool3x3s2_s32(ptr(x), ptr(out), n, h, w, nb * 32, oh, ow, _stream()), "hn_maxpool3x3s2_s32")
        return out
    _req(x, name="x")
    n, h, w, c = x.shape
    oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    if out is None:
        out = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
    check(lib.hn_maxpool3x3s2_nhwc_f32(ptr(x), ptr(out), n, h, w, c, oh, ow, _stream()), "hn_maxpool3x3s2_nhwc_f32")
    return out


def groupnorm_affine(x, gamma, beta, groups=32, eps=1e-5, scratch=None, scale=None, shift=None):
    """x [N,H,W,C] -> (scale [N,C], shift [N,C]) such that GN(x) = x*scale + shift."""
    lib = _lib.load()
    _req(x, name="x"); _req(gamma, name="gamma"); _req(beta, name="beta")
    n, h, w, c = x.shape
    need = lib.hn_groupnorm_scratch_floats(n, h * w, c, groups)
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty((need,), device=x.device, dtype=torch.float32)
    if scale is None:
        scale = torch.empty((n, c), device=x.device, dtype=torch.float32)
    if shift is None:
        shift = torch.empty((n, c), device=x.device, dtype=torch.float32)
    check(lib.hn_groupnorm_affine_f32(ptr(x), ptr(gamma), ptr(beta), n, h * w, c, groups, eps, ptr(scratch),
                                      ptr(scale), ptr(shift), _stream()), "hn_groupnorm_affine_f32")
    return scale, shift


def gn_rows32_scratch_floats(rows, c):
    return _lib.load().hn_groupnorm_rows32_scratch_floats(int(rows), int(c))


def groupnorm_finalize_rows32(partial, gamma, beta, n, hw, groups=32, eps=1e-5, scale=None, shift=None):
    """Partial sums written by conv2d_nhwc(..., gn_partial=partial) -> (scale [N,C], shift [N,C])."""
    lib = _lib.load()
    _req(partial, name="partial"); _req(gamma, name="gamma"); _req(beta, name="beta")
    c = gamma.numel()
    if partial.numel() < lib.hn_groupnorm_rows32_scratch_floats(n * hw, c):
        raise ValueError("partial buffer too small")
    if scale is None:
        scale = torch.empty((n, c), device=partial.device, dtype=torch.float32)
    if shift is None:
        shift = torch.empty((n, c), device=partial.device, dtype=torch.float32)
    check(lib.hn_groupnorm_finalize_rows32(ptr(partial), ptr(gamma), ptr(beta), n, hw, c, groups, eps, ptr(scale),
                                           ptr(shift), _stream()), "hn_groupnorm_finalize_rows32")
    return scale, shift


def groupnorm_finalize_rows32_levels(partials, gamma, beta, n, hws, groups=32, eps=1e-5):
    """groupnorm_finalize_rows32 for all FPN levels of one tower layer in ONE launch -> [(scale, shift)] per level."""
    lib = _lib.load()
    _req(gamma, name="gamma"); _req(beta, name="beta")
    c = gamma.numel()
    lv = _lib.GnLevels()
    lv.count = len(partials)
    tables = torch.empty((len(partials), 2, n, c), device=gamma.device, dtype=torch.float32)
    for i, (part, hw) in enumerate(zip(partials, hws)):
        _req(part, name="partial")
        if part.numel() < lib.hn_groupnorm_rows32_scratch_floats(n * hw, c):
            raise ValueError("partial buffer too small")
        lv.hw[i], lv.partial[i] = hw, part.data_ptr()
        lv.scale[i], lv.shift[i] = tables[i, 0].data_ptr(), tables[i, 1].data_ptr()
    check(lib.hn_groupnorm_finalize_rows32_levels(C.byref(lv), ptr(gamma), ptr(beta), n, c, groups, eps, _stream()),
          "hn_groupnorm_finalize_rows32_levels")
    return [(tables[i, 0], tables[i, 1]) for i in range(len(partials))]


def to_split_levels(xs, affines, relu=True):
    """to_split(x, scale, shift, relu) for the FPN levels of one tower layer in ONE launch (per-level launches when a
    level is too large for the caches, decided by the library).  xs: fp32 [N,h_l,w_l,C]; affines: [(scale, shift)]."""
    lib = _lib.load()
    n, _, _, c = xs[0].shape
    xstride = _pixel_stride(xs[0], "x")
    a_stride = affines[0][0].stride(0)
    lv = _lib.SplitLevels()
    lv.count = len(xs)
    outs = []
    for i, (x, (scale, shift)) in enumerate(zip(xs, affines)):
        if x.shape[0] != n or x.shape[3] != c or _pixel_stride(x, "x") != xstride or is_split(x):
            raise ValueError("levels must be fp32 NHWC tensors of one batch size, channel count and pixel stride")
        for t in (scale, shift):
            if (not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (n, c) or t.stride(1) != 1
                    or t.stride(0) != a_stride):
                raise ValueError("scale / shift must be fp32 GPU [N, C] tables with equal row stride")
        out = torch.empty((n, x.shape[1], x.shape[2], c // 32, 2, 32), device=x.device, dtype=torch.float16)
        lv.hw[i] = x.shape[1] * x.shape[2]
        lv.x[i], lv.scale[i], lv.shift[i], lv.y16[i] = x.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr()
        outs.append(out)
    check(lib.hn_affine_split_f32_levels(C.byref(lv), 1 if relu else 0, n, c, xstride, a_stride, 2 * c, _stream()),
          "hn_affine_split_f32_levels")
    return outs


def _device_readable(t: torch.Tensor, name: str):
    """A raw input buffer of ingest_raw: device memory of the current GPU, or PINNED host memory (the kernel reads it over
    PCIe itself).  Pageable host memory is not readable by the device: the caller stages it (HandNet.forward_raw does)."""
    if t.is_cuda:
        _check_device(t, name)
    elif not t.is_pinned():
        raise RuntimeError(f"{name}: host memory must be pinned (tensor.pin_memory()) to be read by the ingest kernel")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def ingest_raw(bgr_u8, depth=None, device=None, out_rgb=None, out_depth=None, out_rgbd=None, want_rgbd=False, want_depth=True):
    """The reference caller's host-side conversions as ONE kernel (hn_ingest_u8bgr_u16mm; ros_demo.py:227-231,266-269):
    bgr_u8 uint8 [N,H,W,3] (cv_bridge 'bgr8' frames), depth [N,H,W] uint16 / int16 millimetres (16UC1) or float32 metres
    (32FC1), each on the GPU or in PINNED host memory -> (rgb fp32 [N,3,H,W] in 0..1, depth fp32 [N,1,H,W] metres or None,
    rgbd fp32 [N,4,H,W] or None) on the GPU; bit-identical to `astype(float32) / 255.0` and `/ 1000.0`.
    want_depth=False (with an RGB-D output): the separate depth map is neither allocated nor written."""
    if bgr_u8.dtype != torch.uint8 or bgr_u8.dim() != 4 or bgr_u8.shape[3] != 3:
        raise TypeError(f"bgr_u8: expected uint8 [N,H,W,3], got {bgr_u8.dtype} {tuple(bgr_u8.shape)}")
    _device_readable(bgr_u8, "bgr_u8")
    n, h, w, _ = bgr_u8.shape
    kind = 0
    if depth is not None:
        if depth.dtype in (torch.uint16, torch.int16):
            kind = 1        # (int16: the same bits; 16UC1 is unsigned and the kernel reads it so)
        elif depth.dtype == torch.float32:
            kind = 2
        else:
            raise TypeError(f"depth: expected uint16 / int16 millimetres or float32 metres, got {depth.dtype}")
        if tuple(depth.shape) not in ((n, h, w), (n, 1, h, w)):
            raise ValueError(f"depth: expected [N,H,W] matching the frames, got {tuple(depth.shape)}")
        _device_readable(depth, "depth")
    if device is None:
        device = bgr_u8.device if bgr_u8.is_cuda else torch.device("cuda", _cur_device() if _cur_device else 0)
    if out_rgb is None:
        out_rgb = torch.empty((n, 3, h, w), device=device, dtype=torch.float32)
    if want_rgbd and out_rgbd is None:
        if not kind:
            raise ValueError("the RGB-D tensor needs a depth input")
        out_rgbd = torch.empty((n, 4, h, w), device=device, dtype=torch.float32)
    if kind and out_depth is None and (want_depth or out_rgbd is None):
        out_depth = torch.empty((n, 1, h, w), device=device, dtype=torch.float32)
    for t, nm in ((out_rgb, "out_rgb"), (out_depth, "out_depth"), (out_rgbd, "out_rgbd")):
        if t is not None:
            _req(t, name=nm)
    check(_lib.load().hn_ingest_u8bgr_u16mm(bgr_u8.data_ptr(), depth.data_ptr() if kind else None, kind, ptr(out_rgb),
                                            ptr(out_depth) if (kind and out_depth is not None) else None, ptr(out_rgbd),
                                            n, h, w, _stream()),
          "hn_ingest_u8bgr_u16mm")
    return out_rgb, (out_depth if kind else None), out_rgbd


def fcos_preprocess(images, oh, ow, ph, pw, mean, std, out=None):
    """images [N,3,H,W] fp32 (0..1) -> [N,ph,pw,4] normalized / resized / padded NHWC."""
    lib = _lib.load()
    _req(images, name="images")
    n, c, h, w = images.shape
    if c != 3:
        raise ValueError("images must be [N,3,H,W]")
    if out is None:
        out = torch.empty((n, ph, pw, 4), device=images.device, dtype=torch.float32)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.hn_fcos_preprocess_f32(ptr(images), ptr(out), n, h, w, oh, ow, ph, pw, m, s, _stream()),
          "hn_fcos_preprocess_f32")
    return out


def fcos_preprocess_split(images, oh, ow, ph, pw, mean, std, border=3, out=None):
    """images [N,3,H,W] fp32 (0..1) -> stem image: fp16 [2 (hi, lo), N, ph+2b, pw+2b, 4], zero border b."""
    lib = _lib.load()
    _req(images, name="images")
    n, c, h, w = images.shape
    if c != 3:
        raise ValueError("images must be [N,3,H,W]")
    if out is None:
        out = torch.empty((2, n, ph + 2 * border, pw + 2 * border, 4), device=images.device, dtype=torch.float16)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.hn_fcos_preprocess_split(ptr(images), ptr(out), n, h, w, oh, ow, ph, pw, border, m, s, _stream()),
          "hn_fcos_preprocess_split")
    return out


def fcos_preprocess_list(images, geom, ph, pw, mean, std, split=True, border=3):
    """Differently sized images (torchvision batch_images): images = list of fp32 GPU [3,h_i,w_i] tensors,
    geom = [(h_i, w_i, oh_i, ow_i)] -> the common canvas (stem image when split, else fp32 [N,ph,pw,4])."""
    lib = _lib.load()
    n = len(images)
    if n == 0 or len(geom) != n:
        raise ValueError("need one geometry row per image")
    keep = []
    for i, (img, (h, w, oh, ow)) in enumerate(zip(images, geom)):
        img = _req(img.float().contiguous(), name=f"images[{i}]")
        if tuple(img.shape) != (3, h, w) or oh > ph or ow > pw or min(h, w, oh, ow) <= 0:
            raise ValueError(f"images[{i}]: shape {tuple(img.shape)} does not match its geometry {(h, w, oh, ow)} "
                             f"or exceeds the canvas {(ph, pw)}")
        keep.append(img)
    dev = keep[0].device
    ptrs = torch.tensor([t.data_ptr() for t in keep], dtype=torch.int64).to(dev)
    gtab = torch.tensor([list(g) for g in geom], dtype=torch.int32).to(dev)
    if split:
        out = torch.empty((2, n, ph + 2 * border, pw + 2 * border, 4), device=dev, dtype=torch.float16)
    else:
        out = torch.empty((n, ph, pw, 4), device=dev, dtype=torch.float32)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.hn_fcos_preprocess_list(ptr(ptrs), ptr(gtab), ptr(out), 1 if split else 0, n, ph, pw, border, m, s,
                                      _stream()), "hn_fcos_preprocess_list")
    # the kernel reads `keep` / the tables asynchronously: tie their lifetime to the output
    out._hn_sources = (keep, ptrs, gtab)
    return out


def conv_stem_split(x16, w16, bias, cout, r=7, stride=2, relu=True, out_split=True, algo_cin=3, out=None):
    """Stem conv on the f16x3 kernel.  x16: stem image from fcos_preprocess_split (border = r // 2);
    w16: weights.pack_stem_split(...).w16.  -> [N, oh, ow, cout] fp32 or S32."""
    lib = _lib.load()
    pad = r // 2
    if x16.dim() != 5 or x16.shape[0] != 2 or x16.shape[4] != 4 or x16.dtype != torch.float16 or not x16.is_cuda \
            or not x16.is_contiguous():
        raise ValueError("x16 must be the contiguous fp16 GPU stem image [2, N, H+2p, W+2p, 4]")
    n, hb, wb = x16.shape[1:4]
    ph, pw = hb - 2 * pad, wb - 2 * pad
    oh, ow = (hb - r) // stride + 1, (wb - r) // stride + 1
    if tuple(w16.shape) != (cout, r, 2, 32) or w16.dtype != torch.float16 or not w16.is_cuda or not w16.is_contiguous():
        raise ValueError("w16 must be the fp16 GPU tensor [cout, r, 2, 32] from weights.pack_stem_split")
    if out is None:
        if out_split:
            out = torch.empty((n, oh, ow, cout // 32, 2, 32), device=x16.device, dtype=torch.float16)
        else:
            out = torch.empty((n, oh, ow, cout), device=x16.device, dtype=torch.float32)
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv_stem_f16x3(ptr(x16), n, ph, pw, pad, r, stride, cout, ptr(w16), ptr(bias), 1 if relu else 0,
                                 ptr(out), 1 if is_split(out) else 0, _stream()), "hn_conv_stem_f16x3")
    if prof is not None:
        timer.stop()
        tile = 4 if cout <= 32 else (2 if cout <= 64 else 1)  # HN_TILE_128x32 / 128x64 / 128x128
        prof.append((("f16x3", tile), n * oh * ow * cout * r * r * algo_cin, timer, (n, ph, pw, 4, cout, r, stride, 1),
                     PROFILE_STAGE))
    return out


def conv_stem_pool_split(x16, w16, bias, cout=64, r=7, stride=2, algo_cin=3, out=None):
    """Stem conv + ReLU + 3x3/2 max pooling in ONE kernel (hn_conv_stem_pool_f16x3): stem image -> pooled S32 map
    [N, (oh+1)//2, (ow+1)//2, cout/32, 2, 32]; bit-identical to conv_stem_split(...) followed by maxpool3x3s2_nhwc."""
    lib = _lib.load()
    pad = r // 2
    if x16.dim() != 5 or x16.shape[0] != 2 or x16.shape[4] != 4 or x16.dtype != torch.float16 or not x16.is_cuda \
            or not x16.is_contiguous():
        raise ValueError("x16 must be the contiguous fp16 GPU stem image [2, N, H+2p, W+2p, 4]")
    _check_device(x16, "x16")
    n, hb, wb = x16.shape[1:4]
    ph, pw = hb - 2 * pad, wb - 2 * pad
    oh, ow = (hb - r) // stride + 1, (wb - r) // stride + 1
    poh, pow_ = (oh + 2 - 3) // 2 + 1, (ow + 2 - 3) // 2 + 1
    if tuple(w16.shape) != (cout, r, 2, 32) or w16.dtype != torch.float16 or not w16.is_cuda or not w16.is_contiguous():
        raise ValueError("w16 must be the fp16 GPU tensor [cout, r, 2, 32] from weights.pack_stem_split")
    if out is None:
        out = torch.empty((n, poh, pow_, cout // 32, 2, 32), device=x16.device, dtype=torch.float16)
    prof = CONV_PROFILE
    if prof is not None:
        timer = HipTimer()
        timer.start()
    check(lib.hn_conv_stem_pool_f16x3_terms(ptr(x16), n, ph, pw, pad, r, stride, cout, ptr(w16), ptr(bias), ptr(out),
                                            1 if F16_TERMS == 1 else 3, _stream()), "hn_conv_stem_pool_f16x3_terms")
    if prof is not None:
        timer.stop()
        # algorithmic work = the stem convolution as the reference executes it (the patch halo is overhead, not work)
        prof.append((("f16x3", 8), n * oh * ow * cout * r * r * algo_cin, timer, (n, ph, pw, 4, cout, r, stride, 1), PROFILE_STAGE))
    return out


@dataclass
class Candidates:
    boxes: torch.Tensor
    scores: torch.Tensor
    labels: torch.Tensor
    sides: torch.Tensor
    level: torch.Tensor
    count: torch.Tensor
    point: torch.Tensor = None  # anchor-point index of each candidate (row of the [P, ...] head tensors)


@dataclass
class Detections:
    boxes: torch.Tensor
    scores: torch.Tensor
    labels: torch.Tensor
    sides: torch.Tensor
    level: torch.Tensor
    keep: torch.Tensor
    count: torch.Tensor


def _zero_fields(n, cap, device, widths, zero=True):
    """One zero-filled allocation carved into [n, cap(, w)] fields + an [n] counter (a torch.zeros per field was 14
    fill launches per detector pass: 65 us of the 2.8 ms batch-1 step, profiles/r03_b1_timeline.txt).  Every field is
    4 bytes wide; int fields are int32 views of the fp32 buffer."""
    words = sum(widths) * n * cap + n
    flat = (torch.zeros if zero else torch.empty)((words,), device=device, dtype=torch.float32)
    out, off = [], 0
    for w in widths:
        t = flat[off:off + n * cap * w]
        out.append(t.view(n, cap, w) if w > 1 else t.view(n, cap))
        off += n * cap * w
    out.append(flat[off:off + n])
    return out


def alloc_candidates(n, cap, device) -> Candidates:
    # internal to the detector (rows beyond count[i] are never read; hn_fcos_candidates* writes every count): no fill launch
    b, s, l, sd, lv, pt, cnt = _zero_fields(n, cap, device, [4, 1, 1, 1, 1, 1], zero=False)
    i32 = torch.int32
    return Candidates(b, s, l.view(i32), sd.view(i32), lv.view(i32), cnt.view(i32), pt.view(i32))


def alloc_detections(n, cap, device, zero=True) -> Detections:
    """zero=False: no fill launch -- rows at or beyond count[i] are then undefined (hn_fcos_nms writes every count and the
    rows below it; the engines, which only ever read those, allocate this way: one launch fewer per detector pass)."""
    b, s, l, sd, lv, kp, cnt = _zero_fields(n, cap, device, [4, 1, 1, 1, 1, 1], zero=zero)
    i32 = torch.int32
    return Detections(b, s, l.view(i32), sd.view(i32), lv.view(i32), kp.view(i32), cnt.view(i32))


CANDIDATES_CHUNKED = True   # development switch (forms.apply_env): hn_fcos_candidates_ws vs one workgroup per image


def fcos_candidates(cls_lr, reg_ctr, strides, num_classes, score_thresh=0.7, out: Candidates | None = None):
    """cls_lr[l] [N,h,w,C+2], reg_ctr[l] [N,h,w,5] per level -> ordered candidates (hn_fcos_candidates_ws: the points of
    an image spread over 1024-point workgroups; identical output to the one-workgroup form)."""
    lib = _lib.load()
    lv = FcosLevels()
    lv.num_levels = len(cls_lr)
    n = cls_lr[0].shape[0]
    cap = 0
    for i, (a, b, st) in enumerate(zip(cls_lr, reg_ctr, strides)):
        _req(a, name="cls_lr"); _req(b, name="reg_ctr")
        if a.shape[3] != num_classes + 2 or b.shape[3] != 5 or a.shape[:3] != b.shape[:3]:
            raise ValueError("bad head tensor shapes")
        lv.h[i], lv.w[i], lv.stride[i] = a.shape[1], a.shape[2], int(st)
        lv.cls_lr[i], lv.reg_ctr[i] = a.data_ptr(), b.data_ptr()
        cap += a.shape[1] * a.shape[2]
    if out is None:
        out = alloc_candidates(n, cap, cls_lr[0].device)
    points, cap = cap, out.scores.shape[1]
    if CANDIDATES_CHUNKED:
        nbytes = lib.hn_fcos_candidates_ws_bytes(n, points)
        ws = torch.empty((nbytes // 4,), device=cls_lr[0].device, dtype=torch.int32)
        check(lib.hn_fcos_candidates_ws(C.byref(lv), n, num_classes, score_thresh, ptr(out.boxes), ptr(out.scores),
                                        ptr(out.labels), ptr(out.sides), ptr(out.level),
                                        ptr(out.point) if out.point is not None else None, ptr(out.count), cap,
                                        ptr(ws), nbytes, _stream()), "hn_fcos_candidates_ws")
        return out
    check(lib.hn_fcos_candidates(C.byref(lv), n, num_classes, score_thresh, ptr(out.boxes), ptr(out.scores),
                                 ptr(out.labels), ptr(out.sides), ptr(out.level),
                                 ptr(out.point) if out.point is not None else None, ptr(out.count), cap, _stream()),
          "hn_fcos_candidates")
    return out


def fcos_ext_gather(ext, det: "Detections", cand: Candidates):
    """ext[l] [N,h,w,8] raw (relu(dxdy)[3], contact[5]) per level -> (contacts [N,cap] int32,
    dxdymags [N,cap,3] fp32) for the kept detections (fcos.py:299-320,605-607,631-647)."""
    lib = _lib.load()
    if cand.point is None:
        raise ValueError("candidates were produced without anchor-point indices")
    lv = FcosLevels()
    lv.num_levels = len(ext)
    arr = (C.c_void_p * len(ext))()
    for i, e in enumerate(ext):
        _req(e, name="ext")
        if e.shape[3] != 8:
            raise ValueError("ext tensors must be [N,h,w,8]")
        lv.h[i], lv.w[i], lv.stride[i] = e.shape[1], e.shape[2], 1
        arr[i] = e.data_ptr()
    n, cap = det.scores.shape
    if sum(e.shape[1] * e.shape[2] for e in ext) != cand.scores.shape[1] or cand.scores.shape[1] != cap:
        raise ValueError("ext levels do not match the candidate capacity")
    contacts = torch.zeros((n, cap), device=det.scores.device, dtype=torch.int32)
    dxdymags = torch.zeros((n, cap, 3), device=det.scores.device, dtype=torch.float32)
    check(lib.hn_fcos_ext_gather(C.byref(lv), arr, ptr(det.keep), ptr(cand.point), ptr(det.count), n, cap,
                                 ptr(contacts), ptr(dxdymags), _stream()), "hn_fcos_ext_gather")
    return contacts, dxdymags


def fcos_nms(cand: Candidates, iou_thresh, ratio_h, ratio_w, scratch=None, out: Detections | None = None,
             ratios=None):
    """ratios: optional fp32 GPU [N,2] = (ratio_h, ratio_w) per image (batches of differently sized images);
    otherwise the scalar pair applies to every image."""
    lib = _lib.load()
    n, cap = cand.scores.shape
    _check_device(cand.scores, "candidates")
    need = lib.hn_fcos_nms_scratch_bytes(n, cap)
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty((need,), device=cand.scores.device, dtype=torch.uint8)
    if out is None:
        out = alloc_detections(n, cap, cand.scores.device)
    if ratios is not None:
        _req(ratios, name="ratios")
        if tuple(ratios.shape) != (n, 2):
            raise ValueError("ratios must be [N,2]")
        check(lib.hn_fcos_nms_ratios(ptr(cand.boxes), ptr(cand.scores), ptr(cand.labels), ptr(cand.sides),
                                     ptr(cand.level), ptr(cand.count), n, cap, float(iou_thresh), ptr(ratios),
                                     ptr(scratch), ptr(out.boxes), ptr(out.scores), ptr(out.labels), ptr(out.sides),
                                     ptr(out.level), ptr(out.keep), ptr(out.count), _stream()), "hn_fcos_nms_ratios")
        return out
    check(lib.hn_fcos_nms(ptr(cand.boxes), ptr(cand.scores), ptr(cand.labels), ptr(cand.sides), ptr(cand.level),
                          ptr(cand.count), n, cap, float(iou_thresh), float(ratio_h), float(ratio_w), ptr(scratch),
                          ptr(out.boxes), ptr(out.scores), ptr(out.labels), ptr(out.sides), ptr(out.level),
                          ptr(out.keep), ptr(out.count), _stream()), "hn_fcos_nms")
    return out


def nms(boxes, scores, iou_thresh):
    """torchvision.ops.nms semantics: indices of kept boxes, descending score."""
    lib = _lib.load()
    _req(boxes, name="boxes"); _req(scores, name="scores")
    k = boxes.shape[0]
    if k == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    scratch = torch.empty((lib.hn_fcos_nms_scratch_bytes(1, k),), device=boxes.device, dtype=torch.uint8)
    keep = torch.empty((k,), device=boxes.device, dtype=torch.int32)
    cnt = torch.zeros((1,), device=boxes.device, dtype=torch.int32)
    check(lib.hn_nms(ptr(boxes), ptr(scores), k, float(iou_thresh), ptr(scratch), ptr(keep), ptr(cnt), _stream()),
          "hn_nms")
    return keep[: int(cnt.item())].to(torch.int64)


def crop_resize(det: Detections, hand_label, depth, out_size=176, cpad=4, crop_box=None, has_hand=None, crops=None,
                reorder_bgr=False):
    """depth [N,C,H,W] (C = 1 depth, or 4 RGB-D) -> (crop_box [N,4] int64, has_hand [N] int32,
    crops [N,out,out,cpad] NHWC).  reorder_bgr applies the RGBD channel permutation [2,1,0,3]."""
    lib = _lib.load()
    _req(depth, name="depth")
    n, c, h, w = depth.shape
    if c not in (1, 4):
        raise ValueError("depth must be [N,1,H,W] or [N,4,H,W]")
    cap = det.scores.shape[1]
    dev = depth.device
    if crop_box is None:
        crop_box = torch.empty((n, 4), device=dev, dtype=torch.int64)
    if has_hand is None:
        has_hand = torch.empty((n,), device=dev, dtype=torch.int32)
    if crops is None:
        crops = torch.empty((n, out_size, out_size, cpad), device=dev, dtype=torch.float32)
    check(lib.hn_crop_resize(ptr(det.boxes), ptr(det.labels), ptr(det.count), cap, int(hand_label), ptr(depth), n, c,
                             1 if reorder_bgr else 0, h, w, out_size, cpad, ptr(crop_box), ptr(has_hand), ptr(crops),
                             _stream()), "hn_crop_resize")
    return crop_box, has_hand, crops


def stem_image_nhwc4(x, border=3, out=None, valid=None):
    """fp32 [N,H,W,4] -> stem image fp16 [2 (hi, lo), N, H+2b, W+2b, 4] with a zero border (input of conv_stem_*_split).
    valid [N] int32 (optional): images with valid == 1 that hold a non-finite pixel get valid = 2 (a2j_aggregate then writes
    NaN rows for them, as the reference's network does)."""
    _req(x, name="x")
    n, h, w, c = x.shape
    if c != 4:
        raise ValueError("x must be [N,H,W,4]")
    if out is None:
        out = torch.empty((2, n, h + 2 * border, w + 2 * border, 4), device=x.device, dtype=torch.float16)
    if valid is not None:
        _req(valid, torch.int32, "valid")
    check(_lib.load().hn_stem_image_nhwc4_valid(ptr(x), n, h, w, border, ptr(out), ptr(valid) if valid is not None else None,
                                                _stream()), "hn_stem_image_nhwc4_valid")
    return out


def pack_depth_nhwc(depth, cpad=4, out=None):
    """depth [N,1,H,W] -> [N,H,W,cpad] with depth in channel 0."""
    lib = _lib.load()
    _req(depth, name="depth")
    n, c, h, w = depth.shape
    if c != 1:
        raise ValueError("depth must be [N,1,H,W]")
    if out is None:
        out = torch.empty((n, h, w, cpad), device=depth.device, dtype=torch.float32)
    check(lib.hn_pack_depth_nhwc(ptr(depth), ptr(out), n, h * w, cpad, _stream()), "hn_pack_depth_nhwc")
    return out


def a2j_aggregate(cls, reg, dep, joints=21, stride=16, valid=None, out=None, convert=None):
    """cls/dep [K,fh,fw,16*J], reg [K,fh,fw,16*J*2] -> [K,J,3].
    convert: optional dict -- convert_joints + uvd2xyz in the SAME launch (hn_a2j_aggregate_convert_f32):
      crop_box [K,4] int64 (the boxes the crops were cut with), paras = (fx, fy, cx, cy) or None, crop = 176,
      clamp_keypoints / clamp_box = (H, W): the live caller's clamps (ros_demo.py:279-283),
      image_uvd / xyz_mm: preallocated outputs (optional).
    Then returns (crop_uvd, image_uvd [K,J,3], xyz_mm [K,J,3] or None), bit-identical to convert_joints() on crop_uvd."""
    lib = _lib.load()
    _req(cls, name="cls"); _req(reg, name="reg"); _req(dep, name="dep")
    k, fh, fw, aj = cls.shape
    if aj != 16 * joints or reg.shape[3] != 2 * aj or dep.shape != cls.shape:
        raise ValueError("bad head shapes")
    if out is None:
        out = torch.empty((k, joints, 3), device=cls.device, dtype=torch.float32)
    if k == 0:
        return out
    if valid is not None:
        _req(valid, torch.int32, "valid")
    if convert is not None:
        # the evaluation caller's operands (a2j/a2j.py:339-346): the dataset's fp32 boxes / per-sample intrinsics on the device
        sbox, sparas = convert.get("sample_box"), convert.get("sample_paras")
        box = convert.get("crop_box")
        if (box is None) == (sbox is None):
            raise ValueError("convert needs crop_box (int64, the detector's) or sample_box (fp32, the dataset's), one of them")
        for t, dt, nm in ((box, torch.int64, "crop_box"), (sbox, torch.float32, "sample_box"), (sparas, torch.float32, "sample_paras")):
            if t is not None and tuple(_req(t, dt, nm).shape) != (k, 4):
                raise ValueError(f"{nm} must be [K,4]")
        paras = convert.get("paras")
        if paras is not None and sparas is not None:
            raise ValueError("convert: paras (one camera) or sample_paras (one per sample), not both")
        img = convert.get("image_uvd")
        xyz = convert.get("xyz_mm")
        if img is None:
            img = torch.empty((k, joints, 3), device=cls.device, dtype=torch.float32)
        if xyz is None and (paras is not None or sparas is not None):
            xyz = torch.empty((k, joints, 3), device=cls.device, dtype=torch.float32)
        pp = (C.c_float * 4)(*[float(v) for v in paras]) if paras is not None else None
        opts = None
        cb = convert.get("clamp_box")
        if convert.get("clamp_keypoints") or cb or sbox is not None or sparas is not None:
            opts = _lib.ConvertOpts(1 if convert.get("clamp_keypoints") else 0, int(cb[0]) if cb else 0, int(cb[1]) if cb else 0, 0,
                                    ptr(sbox), ptr(sparas))
        crop = float(convert.get("crop", 176))
        check(lib.hn_a2j_aggregate_convert_f32(ptr(cls), ptr(reg), ptr(dep), ptr(valid), k, fh, fw, joints, stride, ptr(box),
                                               crop, crop, pp, C.byref(opts) if opts is not None else None, ptr(out), ptr(img),
                                               ptr(xyz), _stream()), "hn_a2j_aggregate_convert_f32")
        return out, img, xyz
    check(lib.hn_a2j_aggregate_f32(ptr(cls), ptr(reg), ptr(dep), ptr(valid), k, fh, fw, joints, stride, ptr(out),
                                   _stream()), "hn_a2j_aggregate_f32")
    return out


def joints2d_standardize(image_uvd, valid=None, out=None):
    """image (u,v,d) joints [N,J,3] -> the lifter's input [N,J,2]: per frame and axis (x - mean) / std over the joints
    (hn_joints2d_standardize_f32 = the live caller's bbox / affine / normalisation chain, ros_demo.py:148-157, which reduces
    to exactly this when there is no rotation); rows with valid != 1 are zeros."""
    _req(image_uvd, name="image_uvd")
    n, j, _ = image_uvd.shape
    if out is None:
        out = torch.empty((n, j, 2), device=image_uvd.device, dtype=torch.float32)
    if valid is not None:
        _req(valid, torch.int32, "valid")
    check(_lib.load().hn_joints2d_standardize_f32(ptr(image_uvd), ptr(valid), n, j, ptr(out), _stream()),
          "hn_joints2d_standardize_f32")
    return out


def convert_joints(kp, crop_box, valid=None, paras=None, crop=176, out=None):
    """kp [N,J,3] crop-(u,v,d), crop_box [N,4] int64 -> image (u,v,d), or camera xyz in mm when
    paras = (fx, fy, cx, cy) is given (convert_joints + uvd2xyz of the reference, on the device)."""
    lib = _lib.load()
    _req(kp, name="kp")
    _req(crop_box, torch.int64, "crop_box")
    n, j, _ = kp.shape
    if out is None:
        out = torch.empty_like(kp)
    if valid is not None:
        _req(valid, torch.int32, "valid")
    pp = None
    if paras is not None:
        pp = (C.c_float * 4)(*[float(v) for v in paras])
    check(lib.hn_convert_joints_f32(ptr(kp), ptr(crop_box), ptr(valid), n, j, float(crop), float(crop), pp, ptr(out),
                                    _stream()), "hn_convert_joints_f32")
    return out


def convert_joints_samples(kp, box_f32, sample_paras=None, valid=None, crop=176, want_image=True):
    """The evaluation caller's conversion (A2JModelLightning.test_step, a2j/a2j.py:339-346): kp [N,J,3] crop-(u,v,d), box_f32
    [N,4] fp32 = the dataset's boxes (fractional corners), sample_paras [N,4] fp32 = each sample's (fx, fy, cx, cy), all on the
    device -> (image (u,v,d) or None, camera xyz in mm or None); fp32 in the reference's order (bit-identical to numpy)."""
    _req(kp, name="kp")
    n, j, _ = kp.shape
    if tuple(_req(box_f32, torch.float32, "box_f32").shape) != (n, 4):
        raise ValueError("box_f32 must be [N,4]")
    if sample_paras is not None and tuple(_req(sample_paras, torch.float32, "sample_paras").shape) != (n, 4):
        raise ValueError("sample_paras must be [N,4]")
    if valid is not None:
        _req(valid, torch.int32, "valid")
    if not want_image and sample_paras is None:
        raise ValueError("nothing to compute: no image (u,v,d) wanted and no intrinsics given")
    img = torch.empty_like(kp) if want_image else None
    xyz = torch.empty_like(kp) if sample_paras is not None else None
    check(_lib.load().hn_convert_joints_samples_f32(ptr(kp), ptr(box_f32), ptr(sample_paras), ptr(valid), n, j, float(crop),
                                                    float(crop), ptr(img), ptr(xyz), _stream()), "hn_convert_joints_samples_f32")
    return img, xyz


def pack_records(kp, crop_box, has_hand, rows, rec_bytes, out=None, extras=()):
    """One step's per-frame results -> [rows, rec_bytes] uint8 records (rows >= frames: shard padding is zero rows).
    extras: up to two more [N,J,3] fp32 fields behind the keypoints (image uvd, camera xyz: the wide record)."""
    _req(kp, name="keypoints"); _req(crop_box, torch.int64, "crop_box"); _req(has_hand, torch.int32, "has_hand")
    n, j = kp.shape[0], kp.shape[1]
    if out is None:
        out = torch.empty((rows, rec_bytes), device=kp.device, dtype=torch.uint8)
    extras = [e for e in extras if e is not None]
    if extras:
        for e in extras:
            _req(e, name="extra")
            if e.shape != kp.shape:
                raise ValueError("extra fields must have the keypoints' shape")
        e0, e1 = extras[0], (extras[1] if len(extras) > 1 else None)
        check(_lib.load().hn_pack_records_ex(ptr(kp), ptr(crop_box), ptr(has_hand), n, rows, j, rec_bytes, ptr(e0), ptr(e1),
                                             ptr(out), _stream()), "hn_pack_records_ex")
        return out
    check(_lib.load().hn_pack_records(ptr(kp), ptr(crop_box), ptr(has_hand), n, rows, j, rec_bytes, ptr(out), _stream()),
          "hn_pack_records")
    return out


def unpack_records(rec, joints):
    """[rows, rec_bytes] uint8 records -> (keypoints [rows,J,3], crop_box [rows,4] int64, has_hand [rows] int32,
    valid [rows] int32)."""
    _req(rec, torch.uint8, "records")
    rows, rec_bytes = rec.shape
    dev = rec.device
    kp = torch.empty((rows, joints, 3), device=dev, dtype=torch.float32)
    box = torch.empty((rows, 4), device=dev, dtype=torch.int64)
    has = torch.empty((rows,), device=dev, dtype=torch.int32)
    valid = torch.empty((rows,), device=dev, dtype=torch.int32)
    check(_lib.load().hn_unpack_records(ptr(rec), rows, joints, rec_bytes, ptr(kp), ptr(box), ptr(has), ptr(valid), _stream()),
          "hn_unpack_records")
    return kp, box, has, valid


def nonfinite_count(x, flag=None):
    """Adds the number of non-finite values of x (fp32 GPU) to flag[0] (int32 GPU; fresh zero by default); no sync."""
    _req(x, name="x")
    if flag is None:
        flag = torch.zeros((1,), device=x.device, dtype=torch.int32)
    check(_lib.load().hn_nonfinite_count_f32(ptr(x), x.numel(), ptr(flag), _stream()), "hn_nonfinite_count_f32")
    return flag


def tickets_nonzero() -> int:
    """Split-K ticket counters of the current device that are not zero (hn_debug_tickets_nonzero; synchronises).  0 after any
    sequence of completed launches: the counters are zero at rest."""
    c = C.c_int64(0)
    check(_lib.load().hn_debug_tickets_nonzero(C.byref(c)), "hn_debug_tickets_nonzero")
    return int(c.value)


def set_form(name: str, on=True):
    """Kernel-form switch of the library (hn_set_form; names in include/handnet_hip.h): an older form of a kernel as a
    bit-identity reference or for A/B timing.  Process-wide; the product never sets one (hn_amd/forms.py)."""
    check(_lib.load().hn_set_form(name.encode(), 1 if on else 0), "hn_set_form")


def range_check_enable(on=True):
    """f16x3 range contract (see include/handnet_hip.h): split producers launched from now on flag values that
    cannot be represented as fp16 hi + lo (|v| > 65504 or non-finite)."""
    check(_lib.load().hn_range_check_enable(1 if on else 0), "hn_range_check_enable")


def range_check_bits(reset=True) -> int:
    """OR of the _lib.RANGE_* bits a split producer has set in the LIBRARY's flag block since the last reset (synchronises)."""
    flag = C.c_int32(0)
    check(_lib.load().hn_range_check_fetch(C.byref(flag), 1 if reset else 0, _stream()), "hn_range_check_fetch")
    return int(flag.value)


def range_check_fetch(reset=True) -> bool:
    """True if a split producer met an out-of-range / non-finite value since the last reset (synchronises)."""
    return range_check_bits(reset) != 0


def range_check_bind(block=None):
    """Split producers launched from now on note into `block` (4 zeroed int32 on the GPU) instead of the library's own
    flag words; None unbinds.  Host-side state, read at launch time."""
    check(_lib.load().hn_range_check_bind(ptr(block) if block is not None else None), "hn_range_check_bind")


@contextlib.contextmanager
def range_scope(block=None, on=True):
    """The f16x3 range contract for the launches of ONE step of ONE engine: inside the scope the split producers this HOST
    THREAD launches note into `block` (4 zeroed int32 on the GPU; on=False: nowhere); on exit the thread's previous switch and
    block are back (hn_range_scope_begin / _end: per-thread state in the library, so engines on other threads of the process
    are never redirected or switched off)."""
    lib = _lib.load()
    check(lib.hn_range_scope_begin(ptr(block) if (block is not None and on) else None, 1 if on else 0), "hn_range_scope_begin")
    try:
        yield
    finally:
        check(lib.hn_range_scope_end(), "hn_range_scope_end")


def range_check_collect(block=None, out=None):
    """One tiny launch: the flag words of `block` (None: the library's) -> out [4] int32 (activation, input out of range,
    input non-finite, 0), and cleared.  No synchronisation."""
    if out is None:
        dev = block.device if block is not None else torch.device("cuda", torch.cuda.current_device())
        out = torch.empty((4,), device=dev, dtype=torch.int32)
    check(_lib.load().hn_range_check_collect(ptr(block) if block is not None else None, ptr(out), _stream()),
          "hn_range_check_collect")
    return out


def range_bits(words) -> int:
    """host list of the four collected words -> OR of the _lib.RANGE_* bits"""
    return ((_lib.RANGE_ACTIVATION if words[0] else 0) | (_lib.RANGE_INPUT if words[1] else 0)
            | (_lib.RANGE_INPUT_NONFINITE if words[2] else 0))


class RangeError(RuntimeError):
    """A value left the dynamic range the f16x3 split format supports."""


def clock_sample(micros, out=None, stream=None):
    """Enqueue the one-wave shader-clock sampler (MHz -> out[0], fp32 GPU) on `stream` (default: current)."""
    if out is None:
        out = torch.zeros((1,), device=torch.device("cuda", _cur_device() if _cur_device else 0), dtype=torch.float32)
    st = _stream() if stream is None else stream.cuda_stream
    check(_lib.load().hn_clock_sample(int(micros), ptr(out), st), "hn_clock_sample")
    return out


class HipTimer:
    """HIP-event pair recorded on the current stream through the C ABI (bench.py)."""

    def __init__(self):
        lib = _lib.load()
        self._a, self._b = C.c_void_p(), C.c_void_p()
        check(lib.hn_event_create(C.byref(self._a)), "hn_event_create")
        check(lib.hn_event_create(C.byref(self._b)), "hn_event_create")

    def start(self):
        check(_lib.load().hn_event_record(self._a, _stream()), "hn_event_record")

    def stop(self):
        check(_lib.load().hn_event_record(self._b, _stream()), "hn_event_record")

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        check(_lib.load().hn_event_elapsed_ms(self._a, self._b, C.byref(ms)), "hn_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            lib = _lib.load()
            lib.hn_event_destroy(self._a)
            lib.hn_event_destroy(self._b)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------
# Pose2Mesh lifter: Chebyshev graph convolution helpers (csrc/graph_ops.hip)
# ---------------------------------------------------------------------------------------
@dataclass
class CsrGraph:
    indptr: torch.Tensor   # int32 [V+1]
    indices: torch.Tensor  # int32 [nnz], ascending per row
    values: torch.Tensor   # fp32 [nnz]
    v: int


def csr_graph(L, device) -> CsrGraph:
    """scipy sparse matrix (or anything with .tocsr()) / torch sparse tensor -> device CSR."""
    if torch.is_tensor(L):
        c = L.coalesce() if L.layout == torch.sparse_coo else L.to_sparse_coo().coalesce()
        import scipy.sparse as sp
        idx = c.indices().cpu().numpy()
        L = sp.csr_matrix((c.values().cpu().numpy(), (idx[0], idx[1])), shape=tuple(c.shape))
    m = L.tocsr().astype(np.float32)
    m.sort_indices()
    if m.shape[0] != m.shape[1]:
        raise ValueError("graph Laplacian must be square")
    return CsrGraph(torch.from_numpy(m.indptr.astype(np.int32)).to(device), torch.from_numpy(m.indices.astype(np.int32)).to(device),
                    torch.from_numpy(m.data.astype(np.float32)).to(device), m.shape[0])


def cheby2_graph(L, device) -> CsrGraph:
    """2 L L - I as a device CSR (the second-order Chebyshev polynomial of the graph convolution, formed in fp64 on the host)."""
    import scipy.sparse as sp
    m = L.tocsr().astype(np.float64)
    q = (2.0 * (m @ m) - sp.identity(m.shape[0], dtype=np.float64, format="csr")).tocsr()
    q.sum_duplicates()
    return csr_graph(q, device)


def _csr_struct(g: CsrGraph):
    return _lib.GraphCsr(g.indptr.data_ptr(), g.indices.data_ptr(), g.values.data_ptr(), g.v)


def graph_conv_cheby3(g: CsrGraph, g2: CsrGraph, x, cw, relu=True, xin=None, up=1, out_split=False, out=None):
    """One Chebyshev graph convolution (K = 3) as ONE launch (hn_graph_conv_cheby3_f16x3): x fp32 [B,V,Fin] ->
    act(Linear([x0 | L x0 | (2 L L - I) x0]) + bias) (+ feature-axis interpolation of xin [B,V,Fi], rows repeated `up` times);
    cw: ConvW-like with .w [Fout,1,1,K], .w16, .bias over K = pad32(3 Fin) k-major channels.  -> fp32 [B,V*up,Fout] or, with
    out_split, the S32 operand [B,V*up,1,Fout/32,2,32]."""
    _req(x, name="x")
    b, v, fin = x.shape
    fout, _, _, k = cw.w.shape
    if v != g.v or g2.v != g.v or k != (3 * fin + 31) // 32 * 32:
        raise ValueError("graph / filter bank do not match the input")
    if xin is not None:
        _req(xin, name="xin")
        if tuple(xin.shape[:2]) != (b, v):
            raise ValueError("xin must be [B,V,Fi]")
    if out is not None:
        y = _req(out, torch.float16 if out_split else torch.float32, "out")
        if y.numel() != b * v * up * fout * (2 if out_split else 1):
            raise ValueError("out has the wrong size")
    elif out_split:
        y = torch.empty((b, v * up, 1, fout // 32, 2, 32), device=x.device, dtype=torch.float16)
    else:
        y = torch.empty((b, v * up, fout), device=x.device, dtype=torch.float32)
    wf = getattr(cw, "w_frag", None)       # (the filter bank in MFMA fragment order, ops.fragment_order(cw.w16): optional)
    check(_lib.load().hn_graph_conv_cheby3_f16x3(C.byref(_csr_struct(g)), C.byref(_csr_struct(g2)), ptr(x), b, fin,
                                                 ptr(wf if wf is not None else cw.w16), 1 if wf is not None else 0,
                                                 ptr(cw.bias), fout, 1 if relu else 0, ptr(xin), xin.shape[2] if xin is not None else 0,
                                                 up, ptr(y), 1 if out_split else 0, _stream()), "hn_graph_conv_cheby3_f16x3")
    return y


def fragment_order(w16):
    """Split filter bank [Fout, K/32, 2, 32] fp16 -> the same values in MFMA fragment order [ceil(Fout/16), K/32, 2, 64, 8]
    (hn_graph_conv_cheby3_f16x3 with w_frag = 1): element i of lane l = column 16 nt + (l & 15), channel 32 kt + 8 (l >> 4) + i,
    zero rows behind Fout."""
    fout, kt, _, _ = w16.shape
    nt = (fout + 15) // 16
    w = torch.zeros((nt * 16, kt, 2, 32), dtype=w16.dtype, device=w16.device)
    w[:fout] = w16
    # [nt, col 16, kt, pl, chunk 4, i 8] -> [nt, kt, pl, chunk, col, i]: lane = chunk * 16 + col
    return w.view(nt, 16, kt, 2, 4, 8).permute(0, 2, 3, 4, 1, 5).contiguous().view(nt, kt, 2, 64, 8)


def linear_rows(x, cw, scale=None, shift=None, residual=None, relu=False, n_out=None):
    """x fp32 [M <= 4, Kx] -> fp32 [M, N]: act(W pre(x) + bias (+ residual)) as a matrix-vector product (hn_linear_rows_f16x3);
    cw: ConvW-like 1x1 bank (.w [N,1,1,K], .w16, .bias), K >= Kx zero padded; scale / shift [Kx]: pre-activation affine + ReLU."""
    _req(x, name="x")
    m, kx = x.shape
    n, _, _, k = cw.w.shape
    y = torch.empty((m, n), device=x.device, dtype=torch.float32)
    for t, nm in ((scale, "scale"), (shift, "shift"), (residual, "residual")):
        if t is not None:
            _req(t, name=nm)
    check(_lib.load().hn_linear_rows_f16x3(ptr(x), m, x.stride(0), kx, ptr(scale), ptr(shift), ptr(cw.w16), k, n, ptr(cw.bias),
                                           ptr(residual), residual.stride(0) if residual is not None else 0, 1 if relu else 0,
                                           ptr(y), n, _stream()), "hn_linear_rows_f16x3")
    return y


def mesh_finish(mesh, perm, xyz_mm, valid=None, out=None):
    """mesh [N,V0,3], perm int64 [V] (graph_perm_reverse[:V]), xyz_mm [N,J,3] -> [N,V,3] = ((mesh[:, perm] * 1000 + xyz_mm[:, 0])
    / 1000) * (1, -1, -1): out['mesh'] of ros_demo.py:162,332-337, bit-identical to the numpy float32 arithmetic."""
    _req(mesh, name="mesh"); _req(perm, torch.int64, "perm"); _req(xyz_mm, name="xyz_mm")
    n, v0, _ = mesh.shape
    v = perm.shape[0]
    if out is None:
        out = torch.empty((n, v, 3), device=mesh.device, dtype=torch.float32)
    if valid is not None:
        _req(valid, torch.int32, "valid")
    check(_lib.load().hn_mesh_finish_f32(ptr(mesh), ptr(perm), ptr(xyz_mm), ptr(valid), n, v0, v, xyz_mm.shape[1], ptr(out), _stream()),
          "hn_mesh_finish_f32")
    return out


def pad_split_rows(x, cpad):
    """fp32 [rows, f] -> S32 [rows,1,1,cpad/32,2,32], channels f.. zero (hn_pad_split_rows_f32)."""
    _req(x, name="x")
    rows, f = x.shape
    out = torch.empty((rows, 1, 1, cpad // 32, 2, 32), device=x.device, dtype=torch.float16)
    check(_lib.load().hn_pad_split_rows_f32(ptr(x), rows, f, cpad, ptr(out), _stream()), "hn_pad_split_rows_f32")
    return out


def lifter_combine(pose2d, pose3d, fpad=8):
    """pose2d [B,J,2], pose3d [B,J,3] (a view of padded [B,S >= 3J] rows is fine) fp32 -> [B,J,fpad] = [pose2d | pose3d / 1000 | 0]
    (pose2mesh_net.py:20)."""
    _req(pose2d, name="pose2d")
    b, j, _ = pose2d.shape
    if (not pose3d.is_cuda or pose3d.dtype != torch.float32 or tuple(pose3d.shape) != (b, j, 3) or pose3d.stride(2) != 1
            or pose3d.stride(1) != 3 or pose3d.stride(0) < 3 * j):
        raise ValueError("pose3d must be fp32 GPU [B,J,3] with dense joints (rows may be padded)")
    out = torch.empty((b, j, fpad), device=pose2d.device, dtype=torch.float32)
    check(_lib.load().hn_lifter_combine_f32(ptr(pose2d), ptr(pose3d), b, j, pose3d.stride(0), fpad, ptr(out), _stream()),
          "hn_lifter_combine_f32")
    return out


def spmm_csr(g: CsrGraph, x):
    """x fp32 [B,V,F] -> L x."""
    _req(x, name="x")
    b, v, f = x.shape
    if v != g.v:
        raise ValueError("vertex count mismatch")
    y = torch.empty_like(x)
    check(_lib.load().hn_spmm_csr_f32(ptr(g.indptr), ptr(g.indices), ptr(g.values), v, ptr(x), ptr(y), b, f, _stream()),
          "hn_spmm_csr_f32")
    return y


def cheby3_basis_split(g: CsrGraph, x0, x1):
    """x0, x1 = L x0 (fp32 [B,V,F]) -> S32 operand [B,V,1,Cpad/32,2,32] holding [x0 | x1 | 2 L x1 - x0 | 0]."""
    _req(x0, name="x0"); _req(x1, name="x1")
    b, v, f = x0.shape
    cpad = (3 * f + 31) // 32 * 32
    out = torch.empty((b, v, 1, cpad // 32, 2, 32), device=x0.device, dtype=torch.float16)
    check(_lib.load().hn_cheby3_basis_split(ptr(g.indptr), ptr(g.indices), ptr(g.values), v, ptr(x0), ptr(x1), ptr(out),
                                            b, f, cpad, _stream()), "hn_cheby3_basis_split")
    return out


def feat_interp_add(xin, y, up=1):
    """y [B,V,Fo] + linear interpolation of xin [B,V,Fi] along the feature axis, rows repeated `up` times."""
    _req(xin, name="xin"); _req(y, name="y")
    b, v, fo = y.shape
    if tuple(xin.shape[:2]) != (b, v):
        raise ValueError("shape mismatch")
    out = torch.empty((b, v * up, fo), device=y.device, dtype=torch.float32)
    check(_lib.load().hn_feat_interp_add_f32(ptr(xin), ptr(y), ptr(out), b * v, xin.shape[2], fo, up, _stream()),
          "hn_feat_interp_add_f32")
    return out

"""Torch-facing wrappers over the C ABI: allocate outputs with torch, launch on the
current HIP stream.  PyTorch is only the allocator / stream provider here.

Every function requires CUDA(HIP) tensors and raises if the extension is unavailable;
there is deliberately no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, FcosLevels, check, ptr


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (no CPU fallback in handnet-pipeline_amd)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def conv_out_size(h, w, r, s, stride, pad, dil):
    return ((h + 2 * pad - dil * (r - 1) - 1) // stride + 1,
            (w + 2 * pad - dil * (s - 1) - 1) // stride + 1)


def _pixel_stride(t: torch.Tensor, name: str) -> int:
    """Accept a dense NHWC tensor or a channel slice of one (x[..., c0:c1])."""
    n, h, w, c = t.shape
    ps = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != w * ps or t.stride(0) != h * w * ps or ps < c:
        raise ValueError(f"{name} must be NHWC-dense or a channel slice of an NHWC-dense tensor")
    return ps


def make_conv_desc(n, h, w, cin, cout, r, s, stride=1, pad=0, dil=1, relu_cols=0, res_mode=0,
                   res_h=0, res_w=0, in_affine=0, tile=0, precision=0, in_pix_stride=0,
                   out_pix_stride=0) -> ConvDesc:
    oh, ow = conv_out_size(h, w, r, s, stride, pad, dil)
    return ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, r=r, s=s, stride=stride, pad=pad, dil=dil,
                    oh=oh, ow=ow, relu_cols=relu_cols, res_mode=res_mode, res_h=res_h, res_w=res_w,
                    in_affine=in_affine, tile=tile, precision=precision, stats=0, stats_group=0,
                    in_pix_stride=in_pix_stride, out_pix_stride=out_pix_stride, in_affine_stride=0)


# bench.py's roofline leg: when set to a list, every conv launch is bracketed by HIP events
# on the launch stream and (tile id, algorithmic MACs, timer) is appended.
CONV_PROFILE = None
TILE_NAMES = {1: "128x128", 2: "128x64", 3: "64x64", 4: "128x32", 6: "64x128"}


def conv2d_nhwc(x, w, bias=None, *, stride=1, pad=0, dil=1, relu=False, relu_cols=None,
                residual=None, res_upsample=False, in_scale=None, in_shift=None, out=None, tile=0,
                algo_cin=None, w16=None):
    """x [N,H,W,Cin] fp32, w [Cout,R,S,Cin] fp32 -> y [N,OH,OW,Cout].

    algo_cin: input channels the reference's conv really has when Cin is zero-padded
    (only used for FLOP accounting).
    w16: split-fp16 filter bank (weights.split_f16x3); when given the conv runs on the f16
    MFMA with split operands (hn_conv2d_nhwc_f16x3) instead of the f32 MFMA."""
    lib = _lib.load()
    _req(w, name="w")
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("x must be an fp32 GPU tensor (no CPU fallback in handnet-pipeline_amd)")
    xs = _pixel_stride(x, "x")
    n, h, wd, cin = x.shape
    cout, r, s, cin_w = w.shape
    if cin_w != cin:
        raise ValueError(f"weight Cin {cin_w} != input Cin {cin}")
    rc = cout if relu else 0
    if relu_cols is not None:
        rc = relu_cols
    res_mode, rh, rw = 0, 0, 0
    if residual is not None:
        _req(residual, name="residual")
        if res_upsample:
            res_mode, rh, rw = 2, residual.shape[1], residual.shape[2]
        else:
            res_mode = 1
    d = make_conv_desc(n, h, wd, cin, cout, r, s, stride, pad, dil, rc, res_mode, rh, rw,
                       1 if in_scale is not None else 0, tile, in_pix_stride=0 if xs == cin else xs)
    if out is None:
        out = torch.empty((n, d.oh, d.ow, cout), device=x.device, dtype=torch.float32)
    else:
        if tuple(out.shape) != (n, d.oh, d.ow, cout) or not out.is_cuda or out.dtype != torch.float32:
            raise ValueError("out has the wrong shape / dtype / device")
        ys = _pixel_stride(out, "out")
        d.out_pix_stride = 0 if ys == cout else ys
    if res_mode == 1 and tuple(residual.shape) != tuple(out.shape):
        raise ValueError("residual shape mismatch")
    if res_mode == 2 and (residual.shape[0] != n or residual.shape[3] != cout):
        raise ValueError("residual shape mismatch")
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
    if w16 is not None:
        if (not w16.is_cuda or w16.dtype != torch.float16 or not w16.is_contiguous()
                or w16.numel() != 2 * w.numel()):
            raise ValueError("w16 must be the contiguous fp16 GPU tensor produced by weights.split_f16x3(w)")
        check(lib.hn_conv2d_nhwc_f16x3(C.byref(d), ptr(x), ptr(w16), ptr(bias), ptr(residual), ptr(in_scale),
                                       ptr(in_shift), ptr(out), _stream()), "hn_conv2d_nhwc_f16x3")
    else:
        check(lib.hn_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(in_scale),
                                     ptr(in_shift), ptr(out), _stream()), "hn_conv2d_nhwc_f32")
    if prof is not None:
        timer.stop()
        macs = n * d.oh * d.ow * cout * r * s * (algo_cin or cin)
        if w16 is not None:
            kind = ("f16x3", lib.hn_conv2d_f16x3_pick_tile(C.byref(d)))
        else:
            kind = ("f32", lib.hn_conv2d_pick_tile(C.byref(d)))
        prof.append((kind, macs, timer, (n, h, wd, cin, cout, r, stride, dil)))
    return out


def maxpool3x3s2_nhwc(x, out=None):
    lib = _lib.load()
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


@dataclass
class Candidates:
    boxes: torch.Tensor
    scores: torch.Tensor
    labels: torch.Tensor
    sides: torch.Tensor
    level: torch.Tensor
    count: torch.Tensor


@dataclass
class Detections:
    boxes: torch.Tensor
    scores: torch.Tensor
    labels: torch.Tensor
    sides: torch.Tensor
    level: torch.Tensor
    keep: torch.Tensor
    count: torch.Tensor


def alloc_candidates(n, cap, device) -> Candidates:
    i32 = dict(device=device, dtype=torch.int32)
    return Candidates(torch.zeros((n, cap, 4), device=device), torch.zeros((n, cap), device=device),
                      torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32),
                      torch.zeros((n,), **i32))


def alloc_detections(n, cap, device) -> Detections:
    i32 = dict(device=device, dtype=torch.int32)
    return Detections(torch.zeros((n, cap, 4), device=device), torch.zeros((n, cap), device=device),
                      torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32),
                      torch.zeros((n, cap), **i32), torch.zeros((n,), **i32))


def fcos_candidates(cls_lr, reg_ctr, strides, num_classes, score_thresh=0.7, out: Candidates | None = None):
    """cls_lr[l] [N,h,w,C+2], reg_ctr[l] [N,h,w,5] per level -> ordered candidates."""
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
    cap = out.scores.shape[1]
    check(lib.hn_fcos_candidates(C.byref(lv), n, num_classes, score_thresh, ptr(out.boxes), ptr(out.scores),
                                 ptr(out.labels), ptr(out.sides), ptr(out.level), ptr(out.count), cap, _stream()),
          "hn_fcos_candidates")
    return out


def fcos_nms(cand: Candidates, iou_thresh, ratio_h, ratio_w, scratch=None, out: Detections | None = None):
    lib = _lib.load()
    n, cap = cand.scores.shape
    need = lib.hn_fcos_nms_scratch_bytes(n, cap)
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty((need,), device=cand.scores.device, dtype=torch.uint8)
    if out is None:
        out = alloc_detections(n, cap, cand.scores.device)
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


def crop_resize(det: Detections, hand_label, depth, out_size=176, cpad=4, crop_box=None, has_hand=None, crops=None):
    """depth [N,1,H,W] -> (crop_box [N,4] int64, has_hand [N] int32, crops [N,out,out,cpad])."""
    lib = _lib.load()
    _req(depth, name="depth")
    n, c, h, w = depth.shape
    if c != 1:
        raise ValueError("depth must be [N,1,H,W]")
    cap = det.scores.shape[1]
    dev = depth.device
    if crop_box is None:
        crop_box = torch.empty((n, 4), device=dev, dtype=torch.int64)
    if has_hand is None:
        has_hand = torch.empty((n,), device=dev, dtype=torch.int32)
    if crops is None:
        crops = torch.empty((n, out_size, out_size, cpad), device=dev, dtype=torch.float32)
    check(lib.hn_crop_resize(ptr(det.boxes), ptr(det.labels), ptr(det.count), cap, int(hand_label), ptr(depth), n, h, w,
                             out_size, cpad, ptr(crop_box), ptr(has_hand), ptr(crops), _stream()), "hn_crop_resize")
    return crop_box, has_hand, crops


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


def a2j_aggregate(cls, reg, dep, joints=21, stride=16, valid=None, out=None):
    """cls/dep [K,fh,fw,16*J], reg [K,fh,fw,16*J*2] -> [K,J,3]."""
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
    check(lib.hn_a2j_aggregate_f32(ptr(cls), ptr(reg), ptr(dep), ptr(valid), k, fh, fw, joints, stride, ptr(out),
                                   _stream()), "hn_a2j_aggregate_f32")
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

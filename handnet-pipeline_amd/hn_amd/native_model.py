"""ctypes binding of the MODEL-level C ABI (include/handnet_hip.h: hn_create / hn_load_weight / hn_finalize /
hn_fcos_forward / hn_a2j_forward / hn_handnet_forward / hn_destroy).

The layer graphs live in C++ (csrc/model.hip) and issue the same launches as the Python engines, so this class and
`HandNetEngine` return bit-identical tensors (tests/test_model_abi_gpu.py); it exists to exercise that ABI from the
test-suite and as the template for a non-Python host.  torch is only the allocator of inputs / outputs here.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr
from .ops import _stream, on_device


class NativeModel:
    def __init__(self, fcos_sd=None, a2j_sd=None, num_classes=3, num_joints=21, rgbd=False, device="cuda",
                 min_size=0, max_size=0, ext=False, precision="f16x3", image_mean=None, image_std=None):
        """precision: "f16x3" (default), "f16x1" (throughput mode) or "f32" (hn_model_config.precision = HN_PRECISION_F32: the
        exact f32-MFMA kernels, the reference's own arithmetic); image_mean / image_std: the FCOS transform's normalisation."""
        if fcos_sd is None and a2j_sd is None:
            raise ValueError("give a FCOS and / or an A2J state_dict (reference layouts, SURVEY A.6)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("NativeModel runs on the GPU only (no CPU fallback)")
        self.lib = _lib.load()
        self.num_classes, self.num_joints, self.rgbd = num_classes, num_joints, rgbd
        parts = (_lib.MODEL_FCOS if fcos_sd is not None else 0) | (_lib.MODEL_A2J if a2j_sd is not None else 0)
        cfg = _lib.ModelConfig(parts=parts, num_classes=num_classes, num_joints=num_joints, rgbd=1 if rgbd else 0,
                               min_size=min_size, max_size=max_size, ext=1 if ext else 0,
                               f16_terms={"f16x3": 3, "f16x1": 1, "f32": 0}[precision],
                               precision=_lib.PRECISION_F32 if precision == "f32" else _lib.PRECISION_SPLIT)
        if (image_mean is None) != (image_std is None):
            raise ValueError("give image_mean and image_std together")
        if image_std is not None:
            cfg.image_mean = (C.c_float * 3)(*[float(v) for v in image_mean])
            cfg.image_std = (C.c_float * 3)(*[float(v) for v in image_std])
        h = C.c_void_p()
        check(self.lib.hn_create(C.byref(cfg), C.byref(h)), "hn_create")
        self._h = h
        try:
            with on_device(self.device):
                for sd in (fcos_sd, a2j_sd):
                    for name, t in (sd or {}).items():
                        if not torch.is_floating_point(t):
                            continue                       # num_batches_tracked and friends
                        t = t.detach().to("cpu", torch.float32).contiguous()
                        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
                        check(self.lib.hn_load_weight(self._h, name.encode(), t.data_ptr(), shape, t.dim()),
                              "hn_load_weight")
                check(self.lib.hn_finalize(self._h), "hn_finalize")
        except Exception:
            self.close()
            raise

    def close(self):
        if getattr(self, "_h", None):
            self.lib.hn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fcos_capacity(self, h, w) -> int:
        return int(self.lib.hn_fcos_capacity(self._h, h, w))

    def fcos(self, rgb):
        """rgb [N,3,H,W] fp32 GPU -> (boxes [N,cap,4], scores, labels, sides, level [N,cap], count [N])"""
        rgb = rgb.float().contiguous()
        n, _, h, w = rgb.shape
        cap = self.fcos_capacity(h, w)
        with on_device(self.device):
            i32 = dict(device=self.device, dtype=torch.int32)
            out = (torch.zeros((n, cap, 4), device=self.device), torch.zeros((n, cap), device=self.device),
                   torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32),
                   torch.zeros((n,), **i32))
            check(self.lib.hn_fcos_forward(self._h, ptr(rgb), n, h, w, *[ptr(t) for t in out], cap, _stream()),
                  "hn_fcos_forward")
        return out

    def fcos_list(self, images):
        """A list of fp32 GPU [3,h_i,w_i] images of different sizes (torchvision batch_images, fcos.py:702-709) -> the
        same tuple as fcos(), boxes rescaled per image (hn_fcos_forward_list)."""
        imgs = [i.float().contiguous() for i in images]
        k = len(imgs)
        hs = (C.c_int32 * k)(*[int(i.shape[1]) for i in imgs])
        ws = (C.c_int32 * k)(*[int(i.shape[2]) for i in imgs])
        ptrs = (C.c_void_p * k)(*[i.data_ptr() for i in imgs])
        cap = int(self.lib.hn_fcos_capacity_list(self._h, hs, ws, k))
        with on_device(self.device):
            i32 = dict(device=self.device, dtype=torch.int32)
            out = (torch.zeros((k, cap, 4), device=self.device), torch.zeros((k, cap), device=self.device),
                   torch.zeros((k, cap), **i32), torch.zeros((k, cap), **i32), torch.zeros((k, cap), **i32),
                   torch.zeros((k,), **i32))
            check(self.lib.hn_fcos_forward_list(self._h, ptrs, hs, ws, k, *[ptr(t) for t in out], cap, _stream()),
                  "hn_fcos_forward_list")
            torch.cuda.current_stream().synchronize()   # the graph reads `imgs` asynchronously: keep them alive until it has
        return out

    def fcos_ext(self, rgb):
        """ext=True detector: fcos() outputs + (contacts [N,cap] int32, dxdymags [N,cap,3])"""
        rgb = rgb.float().contiguous()
        n, _, h, w = rgb.shape
        cap = self.fcos_capacity(h, w)
        with on_device(self.device):
            i32 = dict(device=self.device, dtype=torch.int32)
            out = (torch.zeros((n, cap, 4), device=self.device), torch.zeros((n, cap), device=self.device),
                   torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32), torch.zeros((n, cap), **i32),
                   torch.zeros((n,), **i32), torch.zeros((n, cap), **i32), torch.zeros((n, cap, 3), device=self.device))
            check(self.lib.hn_fcos_forward_ext(self._h, ptr(rgb), n, h, w, *[ptr(t) for t in out], cap, _stream()),
                  "hn_fcos_forward_ext")
        return out

    def a2j(self, crops, valid=None):
        """crops [K,1,H,W] fp32 GPU -> keypoints [K,J,3] on the GPU"""
        crops = crops.float().contiguous()
        k, c, h, w = crops.shape
        if c != 1:
            raise ValueError("expected [K,1,H,W] depth crops")
        with on_device(self.device):
            kp = torch.empty((k, self.num_joints, 3), device=self.device)
            check(self.lib.hn_a2j_forward(self._h, ptr(crops), k, h, w, ptr(valid), ptr(kp), _stream()), "hn_a2j_forward")
        return kp

    def handnet(self, rgb, depth):
        """rgb [N,3,H,W], depth [N,1|4,H,W] fp32 GPU -> (keypoints [N,J,3], crop_box [N,4] int64, has_hand [N] int32)"""
        rgb, depth = rgb.float().contiguous(), depth.float().contiguous()
        n, _, h, w = rgb.shape
        with on_device(self.device):
            kp = torch.empty((n, self.num_joints, 3), device=self.device)
            box = torch.empty((n, 4), device=self.device, dtype=torch.int64)
            has = torch.empty((n,), device=self.device, dtype=torch.int32)
            check(self.lib.hn_handnet_forward(self._h, ptr(rgb), ptr(depth), n, h, w, ptr(kp), ptr(box), ptr(has), _stream()),
                  "hn_handnet_forward")
        return kp, box, has

    def handnet_xyz(self, rgb, depth, paras=None, clamp=False):
        """hn_handnet_forward_xyz: handnet() + the aggregation epilogue's image (u,v,d) and -- with paras = (fx, fy, cx, cy) --
        camera xyz in mm -> (keypoints, image_uvd, xyz_mm or None, crop_box, has_hand); clamp: the live caller's clamps
        (ros_demo.py:279-283) on what is converted."""
        from . import _lib
        rgb, depth = rgb.float().contiguous(), depth.float().contiguous()
        n, _, h, w = rgb.shape
        with on_device(self.device):
            kp = torch.empty((n, self.num_joints, 3), device=self.device)
            img = torch.empty((n, self.num_joints, 3), device=self.device)
            xyz = torch.empty((n, self.num_joints, 3), device=self.device) if paras is not None else None
            box = torch.empty((n, 4), device=self.device, dtype=torch.int64)
            has = torch.empty((n,), device=self.device, dtype=torch.int32)
            pp = (C.c_float * 4)(*[float(v) for v in paras]) if paras is not None else None
            opts = _lib.ConvertOpts(1, h, w) if clamp else None
            check(self.lib.hn_handnet_forward_xyz(self._h, ptr(rgb), ptr(depth), n, h, w, pp, C.byref(opts) if opts is not None else None,
                                                  ptr(kp), ptr(img), ptr(xyz), ptr(box), ptr(has), _stream()), "hn_handnet_forward_xyz")
        return kp, img, xyz, box, has


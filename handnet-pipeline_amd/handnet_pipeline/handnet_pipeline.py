"""Drop-in `handnet_pipeline.handnet_pipeline.HandNet` (alias `HandNetPipeline`) on MI355X.

Constructor and call contract of handnet_pipeline/handnet_pipeline.py:38-116:

    net = HandNet(args, reload_detector=True, num_classes=3, reload_a2j=True, RGBD=False).cuda().eval()
    keypoints, depth_batch, crops = net(images, depth_images=depth)        # ros_demo.py:270,388

  keypoints   FloatTensor [N,21,3] on the CPU (zero rows for frames without a hand)
  depth_batch [K,1,176,176] on the model device, K = frames with a hand
  crops       [K,4] int64 on the model device (padded, clamped x1,y1,x2,y2)
  no frame with a hand: (zeros[N,21,3], zeros_like(depth_images), zeros[N,4] float32 on the CPU)
  is_detect or is_3D: returns None (the reference has no such branch either).

Documented deviations: a batch that MIXES frames with and without a hand raises in the
reference (torch.stack of a list containing None, :82,111) and an empty crop slice reuses
the previous frame's crop (:100-105); here such frames simply count as "no hand".
"""
from __future__ import annotations

import os

import torch

from a2j.a2j import A2JModel
from fcos_utils.fcos import FCOS
from a2j.a2j import A2JModelLightning
from hn_amd import ops
from hn_amd.pipeline import HandNetEngine, check_range_contract
from hn_amd.state import EngineOwner


def load_pretrained_fcos(args, reload_detector=False, num_classes=2):
    detector = FCOS(num_classes=num_classes, ext=False, nms_thresh=0.5)
    if reload_detector:
        checkpoint = torch.load(args.pretrained_fcos, map_location="cpu")
        detector.load_state_dict(checkpoint["model"], strict=False)
    for p in detector.parameters():
        p.requires_grad = False
    return detector


def load_pretrained_a2j(args, reload_a2j=False, RGBD=False):
    """handnet_pipeline.py:25-36: RGBD or a path containing 'ckpt' -> Lightning checkpoint through
    A2JModelLightning.load_from_checkpoint (the file's hyper_parameters decide the stem width); else A2JModel +
    optional {"model": sd}."""
    if RGBD or "ckpt" in str(args.pretrained_a2j):
        return A2JModelLightning.load_from_checkpoint(args.pretrained_a2j).eval()
    a2j = A2JModel(21, crop_height=176, crop_width=176, is_RGBD=False)
    if reload_a2j:
        checkpoint = torch.load(args.pretrained_a2j, map_location="cpu")
        a2j.load_state_dict(checkpoint["model"], strict=False)
    for p in a2j.parameters():
        p.requires_grad = False
    return a2j


class HandNet(EngineOwner):
    """End-to-End HandNet: FCOS hand detector -> depth crop -> A2J keypoints."""

    def __init__(self, args, reload_detector: bool = False, num_classes: int = 2, reload_a2j: bool = False,
                 RGBD: bool = False):
        super().__init__()
        self.detector = load_pretrained_fcos(args, reload_detector, num_classes)
        self.detector.eval()
        self.a2j = load_pretrained_a2j(args, reload_a2j, RGBD)
        # the reference keeps the caller's flag (handnet_pipeline.py:55); a Lightning checkpoint knows its own stem
        self.RGBD = bool(self.a2j.rgbd) if isinstance(self.a2j, A2JModelLightning) else bool(RGBD)
        self.num_classes = num_classes
        self._auto_graph_allowed = os.environ.get("HN_AUTO_GRAPH", "1") != "0"

    def engine(self) -> HandNetEngine:
        # the per-call path: the engines stand and belong to the sub-modules' CURRENT state -- load_state_dict() and every
        # device / dtype move (nn.Module._apply) reset a sub-module's engine (hn_amd/state.py), which this test sees.  (Walking
        # the three parameter trees for their device on every call was 20 us of the ~90 us the call costs beyond its GPU time.)
        eng = self._engine
        if (eng is not None and getattr(self.detector, "_engine", None) is eng.fcos
                and getattr(self.a2j, "_engine", None) is eng.a2j):
            return eng
        self._require_gpu()
        fcos, a2j = self.detector.engine(), self.a2j.engine()
        # the sub-modules rebuild their engines when THEIR weights change (net.detector.load_state_dict(...)):
        # never keep running a holder of stale ones
        if self._engine is None or self._engine.fcos is not fcos or self._engine.a2j is not a2j:
            self._engine = HandNetEngine(fcos, a2j, self.num_classes)
            if getattr(self, "_convert_cfg", None) is not None:
                self._engine.set_convert(*self._convert_cfg)
        return self._engine

    def set_convert(self, paras=None, clamp: bool = False, on: bool = True):
        """What the reference's caller does with every result (ros_demo.py:279-290,329-330: clamp, convert_joints to image
        (u,v,d), uvd2xyz to camera millimetres) as part of the step: the aggregation's own launch writes them and the call's one
        device -> host record carries them.  After each forward(): `net.last_converted` = {"image_uvd": [N,21,3] CPU,
        "xyz_mm": [N,21,3] CPU or None (no intrinsics)}; forward()'s tuple itself is unchanged.  paras = (fx, fy, cx, cy)."""
        self._convert_cfg = (paras, bool(clamp)) if on else None
        if self._engine is not None:
            self._engine.set_convert(paras, clamp, on)
        self.last_converted = None
        return self

    def live(self, lifter, paras, clamp: bool = True, perm_reverse=None):
        """The live caller's chain as ONE step (hn_amd.live.LiveHandEngine; ros_demo.py:270-290,329-337): this network, the
        caller's clamp + convert_joints (in the aggregation's epilogue), the lifter's input, Pose2Mesh, one device -> host copy.
        lifter: the drop-in `models.pose2mesh_net.get_model(...)` module (on the GPU) or a Pose2MeshEngine; paras = (fx, fy,
        cx, cy); perm_reverse = graph_perm_reverse[:V]: the step then also does ros_demo.py:162,332-337 and hands over out['mesh'].
        The returned engine owns this network's step from then on (forward() of this module keeps working and
        carries the converted joints: set_convert)."""
        from hn_amd.live import LiveHandEngine
        self._convert_cfg = (tuple(paras), bool(clamp))
        return LiveHandEngine(self.engine(), lifter.engine() if hasattr(lifter, "engine") else lifter, paras, clamp, perm_reverse)

    # forward() switches ITSELF to hipGraph replay once the same input shapes have come in a few times in a row -- the live
    # caller's case (ros_demo.py:270-273: one 640x480 frame per call, ~150 dependent launches whose host cost is 8 % of the
    # call; at batch 32 the replay saves the ~0.3 ms the GPU idles while Python issues the first launches after the sync).
    # forward() hands out fresh tensors (keypoints on the CPU, copies of the crops), so replaying into captured buffers is
    # invisible to the caller; enable_graph(False) turns it off, enable_graph(True) forces it from the first call and also for
    # forward_device().
    AUTO_GRAPH_CALLS = 3        # same-shape calls in a row before forward() captures
    AUTO_GRAPH_MAX_SHAPES = 4   # captured steps kept at a time (each holds its own static activation pool): one more input
    #                             shape EVICTS the least recently used capture (a batch-size-sweeping caller stays bounded)
    # HN_AUTO_GRAPH=0 in the environment keeps forward() eager; a capture that FAILS (no memory for the static pool, a
    # capture-unsafe call from another thread of the host) is not an error of the call: forward() runs that call eagerly and
    # never tries again (self._auto_graph_allowed = False), see _forward_auto.
    # A SPARSE stream (a hand in fewer than half of the frames of a batch of >= 8: forward() sees the flags on the CPU anyway)
    # stays eager, because the engine then runs A2J on the frames with a hand only and that path is data dependent.

    def enable_graph(self, on=True):
        """hipGraph replay: the first call with a given input shape captures the whole step, later calls copy the inputs
        into the captured buffers and replay (no per-launch host cost; the launch sequence is static by construction).
        on=True: always (results of forward_device() then alias the captured output buffers and are overwritten by the next
        call; forward() returns fresh tensors either way); on=False: never; on=None: the default -- forward() decides by
        itself (see AUTO_GRAPH_*), forward_device() stays eager."""
        self.use_graph = None if on is None else bool(on)
        if on:
            self._auto_graph_allowed = True
        return self

    def _auto_graph(self, image_shape, depth_shape, on_gpu=True) -> bool:
        """Whether this call of forward() should run as a graph replay (capturing first if need be)."""
        eng = self.engine()
        if not self._auto_graph_allowed or eng.check_range or getattr(self, "_last_sparse", False) or not on_gpu:
            return False
        if eng.has_graph(image_shape, depth_shape, to_host=True):
            return True
        key = (tuple(image_shape), tuple(depth_shape))
        if key == getattr(self, "_streak_key", None):
            self._streak += 1
        else:
            self._streak_key, self._streak = key, 1
        return self._streak > self.AUTO_GRAPH_CALLS

    def forward_device(self, images, depth_images, _graph=None, _to_host=False):
        """Sync-free variant: returns hn_amd.pipeline.HandNetOutput with everything on the GPU."""
        batch = images if torch.is_tensor(images) else torch.stack([i.float() for i in images])
        if getattr(self, "use_graph", None) if _graph is None else _graph:
            batch, depth = batch.float().contiguous(), depth_images.float().contiguous()
            run, s_img, s_dep, out = self.engine().graphed(batch, depth, to_host=_to_host, limit=self.AUTO_GRAPH_MAX_SHAPES)
            s_img.copy_(batch)
            s_dep.copy_(depth)
            run()
            return out
        return self.engine().forward_device(batch, depth_images, to_host=_to_host)

    def _forward_auto(self, images, depth_images, n):
        """forward() in its default mode: replay when a captured step fits, capture when the shapes have repeated, else eager."""
        eng = self.engine()
        if (self._auto_graph_allowed and not torch.is_tensor(images) and n and depth_images.is_cuda
                and depth_images.dtype == torch.float32 and all(i.dtype == torch.float32 and i.is_cuda for i in images)
                and not eng.check_range and not getattr(self, "_last_sparse", False)):
            out = eng.replay_frames(images, depth_images, to_host=True)
            if out is not None:
                return out
        batch = images if torch.is_tensor(images) else torch.stack([i.float() for i in images])
        if self._auto_graph(batch.shape, depth_images.shape, batch.is_cuda and depth_images.is_cuda):
            try:
                return self.forward_device(batch, depth_images, _graph=True, _to_host=True)
            except ops.RangeError:
                raise
            except Exception as e:  # noqa: BLE001 -- whatever made the capture fail, this call would succeed eagerly
                self._capture_failed(e)
        return self.forward_device(batch, depth_images, _graph=False, _to_host=True)

    def _capture_failed(self, e):
        import warnings
        self._auto_graph_allowed = False
        torch.cuda.synchronize()
        warnings.warn(f"HandNet: automatic hipGraph capture failed ({type(e).__name__}: {e}); "
                      "staying eager from now on (enable_graph(True) forces a new attempt)")

    def forward(self, images, depth_images=None, is_3D: bool = False, is_detect: bool = False):
        if is_detect or is_3D:
            return None
        if depth_images is None:
            raise ValueError("depth_images is required for the ensemble inference branch")
        n = len(images)
        mode = getattr(self, "use_graph", None)
        if mode is None and torch.is_tensor(depth_images):
            out = self._forward_auto(images, depth_images, n)
        else:
            out = self.forward_device(images, depth_images, _to_host=True)
        return self._finish(out, n, depth_images)

    def forward_raw(self, bgr_u8, depth_raw, is_3D: bool = False, is_detect: bool = False):
        """The reference caller's ingest AND its network call in one (ros_demo.py:227-231,266-273): bgr_u8 = the cv_bridge
        'bgr8' frames, uint8 [N,H,W,3] (or one [H,W,3] frame); depth_raw = 16UC1 millimetres as uint16 [N,H,W] or 32FC1 metres
        as float32 -- numpy arrays or torch tensors, on the host (pinned memory is read in place over PCIe, pageable memory is
        staged once) or on the GPU.  One kernel does `astype(float32) / 255.0`, BGR -> RGB, HWC -> CHW and `/ 1000.0` (bit-identical
        to the host arithmetic; 1.5 MB per frame cross PCIe instead of 4.9 MB) and the step runs exactly as forward() runs it.
        Returns forward()'s tuple; its no-hand placeholder `zeros_like(depth_images)` has the fp32 [N,1|4,H,W] shape forward()
        would have been given."""
        if is_detect or is_3D:
            return None
        bgr = torch.as_tensor(bgr_u8)
        dep = torch.as_tensor(depth_raw)
        if bgr.dim() == 3:
            bgr = bgr.unsqueeze(0)
        if dep.dim() == 2:
            dep = dep.unsqueeze(0)
        if dep.dim() == 4 and dep.shape[1] == 1:
            dep = dep[:, 0]
        n, h, w = bgr.shape[0], bgr.shape[1], bgr.shape[2]
        eng = self.engine()
        shapes = ((n, 3, h, w), (n, 4 if self.RGBD else 1, h, w))
        mode = getattr(self, "use_graph", None)
        graph = bool(mode) if mode is not None else self._auto_graph(*shapes)
        out = None
        if graph:
            try:
                out = eng.forward_raw(bgr, dep, to_host=True, use_graph=True, limit=self.AUTO_GRAPH_MAX_SHAPES)
            except (ops.RangeError, TypeError, ValueError):
                raise
            except Exception as e:  # noqa: BLE001
                if mode:
                    raise
                self._capture_failed(e)
        if out is None:
            out = eng.forward_raw(bgr, dep, to_host=True)
        return self._finish(out, n, None, placeholder_shape=shapes[1])

    def _finish(self, out, n, depth_images, placeholder_shape=None):
        """The reference's return tuple from a step's device results: ONE device -> host copy (and sync) per call -- the
        step's record buffer (keypoints, has-hand flags, crop boxes, range-contract words), which the step has already enqueued
        into pinned memory."""
        from hn_amd.pipeline import read_host_record
        # the usual case -- every frame has a hand -- needs fresh copies of the crop boxes and the depth crops; they are
        # enqueued BEFORE the sync (hidden behind the step instead of trailing it) and thrown away in the other cases
        # (.contiguous() of the permuted / sliced view is a copy: the caller never holds a view of a captured buffer)
        sel = out.crops_nhwc
        crops_all = out.crop_box.clone()
        depth_all = (sel.permute(0, 3, 1, 2) if self.RGBD else sel[..., 0].unsqueeze(1)).contiguous()
        torch.cuda.current_stream(out.keypoints.device).synchronize()
        if out.image_uvd is not None:
            final_results, has, _box, words, more = read_host_record(out.host_record, n, out.keypoints.shape[1], extras=True)
            self.last_converted = {"image_uvd": more[0], "xyz_mm": more[1] if len(more) > 1 else None}
        else:
            final_results, has, _box, words = read_host_record(out.host_record, n, out.keypoints.shape[1])
        mask_cpu = has != 0
        hands = int(mask_cpu.sum())
        self._last_sparse = n >= 8 and hands * 2 < n      # (the engine's own threshold for compaction)
        # the f16x3 range contract, decided on what has just been copied (hn_amd.pipeline.check_range_contract): overflow raises,
        # non-finite depth pixels give NaN rows like the reference
        check_range_contract(final_results, words if out.range_flags is not None else None, depth_images, has_hand=has)
        if hands == 0:  # handnet_pipeline.py:107-108: the crops placeholder is a CPU float tensor
            zeros = torch.zeros_like(depth_images) if depth_images is not None else torch.zeros(
                placeholder_shape, device=out.keypoints.device)
            return torch.zeros((n, 21, 3)), zeros, torch.zeros((n, 4))
        if hands == n:
            return final_results, depth_all, crops_all
        idx = mask_cpu.nonzero().flatten().to(out.crop_box.device)
        return final_results, depth_all.index_select(0, idx), crops_all.index_select(0, idx)


HandNetPipeline = HandNet

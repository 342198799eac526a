"""The live caller's whole chain as ONE captured step (ros_demo.py:270-290,329-337):

    HandNet (FCOS -> crop -> A2J)                                  handnet_pipeline.py:58-116
      -> clamp + convert_joints: image (u,v), camera xyz in mm     ros_demo.py:279-283,329-330 -- the aggregation's epilogue
      -> the lifter's input (bbox / affine / standardisation)      ros_demo.py:148-157        -- hn_joints2d_standardize_f32
      -> Pose2Mesh (PoseNet MLP + Chebyshev graph convolutions)    ros_demo.py:161, pose2mesh/lib/models/*
      -> ONE device -> host copy: the wide per-frame records (crop box, flags, crop uvd, image uvd, xyz) + the mesh vertices

Everything between the frame and the copy is a static launch sequence on the device: one hipGraph, no host round trip between
the pose network and the lifter (the reference copies the keypoints to the CPU, converts them in numpy and uploads the
normalised joints again, per frame).  What stays the caller's: the vertex permutation / camera offset of the final mesh
(`pred_mesh[:, graph_perm_reverse[:V]]`, `mesh * 1000 + joints3d[0]`, ros_demo.py:162,332-337) -- the step hands over what
`model(joint_img)` and `convert_joints` return.

`CropMeshEngine` is the same chain without the detector, for the reference's stand-alone mesh demo (a2j_mesh.py:58-80): dataset
crops + the dataset's float32 boxes + per-sample intrinsics -> A2J -> clip + convert -> lifter input -> Pose2Mesh -> final mesh.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import ops
from .pipeline import HandNetEngine, HandNetOutput, read_host_record, record_bytes
from .pose2mesh_engine import Pose2MeshEngine


def _same_device(a, b) -> bool:
    """"cuda" and "cuda:<current device>" name the same card"""
    a, b = torch.device(a), torch.device(b)
    index = lambda d: d.index if d.index is not None else torch.cuda.current_device()
    return a.type == b.type and (a.type != "cuda" or index(a) == index(b))


@dataclass
class LiveOutput:
    hand: HandNetOutput          # the step's detector / pose results (image_uvd and xyz_mm included), on the device
    pose2d: torch.Tensor         # [N,21,2] the lifter's standardised input
    mesh: torch.Tensor           # [N,V0,3] Pose2Mesh vertices (finest level of the graph hierarchy, coarsening order), or --
    #                              with perm_reverse -- [N,V,3] = out['mesh'] of ros_demo.py:337 (camera frame, original order)
    pose3d: torch.Tensor         # [N,21,3] PoseNet's lifted joints (millimetre scale of the lifter's training set)
    host: torch.Tensor           # pinned uint8: (N + 1) wide records, then the mesh as fp32 -- ONE copy, enqueued by the step
    n: int = 0
    raw_mesh: torch.Tensor = None   # [N,V0,3] the lifter's own output on the device (= mesh without perm_reverse)

    def read(self):
        """After the stream is synchronised: (keypoints, has_hand, crop_box, range words, [image_uvd, xyz_mm], mesh) as fresh CPU
        tensors."""
        rb = record_bytes(3)
        n = self.n
        rec = self.host[: (n + 1) * rb].view(n + 1, rb)
        kp, has, box, words, more = read_host_record(rec, n, extras=True)
        mesh = self.host[(n + 1) * rb:].view(torch.float32).reshape(n, -1, 3).clone()
        return kp, has, box, words, more, mesh


class LiveHandEngine:
    def __init__(self, hand: HandNetEngine, lifter: Pose2MeshEngine, paras, clamp: bool = True, perm_reverse=None):
        """paras = (fx, fy, cx, cy) of the depth camera (ros_demo.py:191-196); clamp: the caller's clamps before the
        conversion (ros_demo.py:279-283).  perm_reverse: graph_perm_reverse[:V] (int64, V = vertices of the real mesh,
        ros_demo.py:162) -- given, the step also does the caller's last three lines (vertex order, camera offset by the first
        joint, y / z negated: ros_demo.py:332-337) and `mesh` of the outputs IS out['mesh'], [N,V,3]; else the lifter's raw
        [N,V0,3] vertices in coarsening order."""
        if not _same_device(hand.device, lifter.device):
            raise ValueError(f"HandNet on {hand.device} but the lifter on {lifter.device}")
        self.hand, self.lifter, self.device = hand, lifter, hand.device
        hand.set_convert(paras=paras, clamp=clamp)
        self.perm = None
        if perm_reverse is not None:
            self.perm = torch.as_tensor(perm_reverse).to(torch.int64).to(self.device).contiguous()
            if int(self.perm.max()) >= lifter.graphs[0].v or int(self.perm.min()) < 0:
                raise ValueError("perm_reverse points outside the lifter's finest graph")
        self.vertices = lifter.graphs[0].v if self.perm is None else int(self.perm.shape[0])
        self._graphs = {}
        self._buffers = {}

    def _out_buffers(self, n, v0):
        key = (n, v0)
        b = self._buffers.get(key)
        if b is None:
            rb = record_bytes(3)
            nbytes = (n + 1) * rb + n * v0 * 12
            with torch.inference_mode(False):
                b = self._buffers[key] = (torch.zeros((nbytes,), dtype=torch.uint8, device=self.device),
                                          torch.zeros((nbytes,), dtype=torch.uint8, pin_memory=True))
        return b

    @ops.device_guarded
    def forward_device(self, images, depth, _buffers=None) -> LiveOutput:
        """images [N,3,H,W] 0..1 (or a list), depth [N,1,H,W] metres on the GPU -> LiveOutput (no sync)."""
        n = len(images)
        v0 = self.vertices
        dev, host = _buffers if _buffers is not None else self._out_buffers(n, v0)
        rb = record_bytes(3)
        rec = dev[: (n + 1) * rb].view(n + 1, rb)
        mesh_buf = dev[(n + 1) * rb:].view(torch.float32).view(n, v0, 3)

        def lift(_kp, image_uvd, xyz, has_hand):
            # (inside the step's range scope: the lifter's split producers note into the step's flag words, which the step's one
            # collect launch hands over -- an overflowing activation of the lifter raises like one of the pose network)
            p2d = ops.joints2d_standardize(image_uvd, valid=has_hand)
            if self.perm is None:
                mesh, pose3d = self.lifter.forward(p2d, mesh_out=mesh_buf)          # the last layer writes into the copy buffer
                return p2d, mesh, pose3d, mesh
            raw, pose3d = self.lifter.forward(p2d)
            return p2d, ops.mesh_finish(raw, self.perm, xyz, valid=has_hand, out=mesh_buf), pose3d, raw
        # the step packs its wide records and its range words straight into `rec`; ONE copy moves records + mesh
        out = self.hand.forward_device(images, depth, _record=(None, rec), _tail=lift)
        p2d, mesh, pose3d, raw = out.tail
        host.copy_(dev, non_blocking=True)
        return LiveOutput(out, p2d, mesh, pose3d, host, n, raw)

    @ops.device_guarded
    def forward_raw(self, bgr_u8, depth_raw) -> LiveOutput:
        """The camera's buffers in, the mesh out: bgr_u8 uint8 [N,H,W,3] (cv_bridge 'bgr8'), depth_raw [N,H,W] uint16 millimetres
        (16UC1) or float32 metres (32FC1), on the GPU or on the host (pinned: read in place; pageable: staged) -- ONE ingest
        kernel writes the captured step's input buffers (ros_demo.py:227-231,266-269) and the live step replays (captured at
        the first call with these shapes).  Returns the capture's static LiveOutput (overwritten by the next call); no sync."""
        staged = []
        bgr, dep = self.hand._device_readable(bgr_u8, staged), self.hand._device_readable(depth_raw, staged)
        n, h, w, _ = bgr.shape
        key = ((n, 3, h, w), (n, 1, h, w))
        if key not in self._graphs:
            rgb, d1, _ = ops.ingest_raw(bgr, dep, device=self.device)
            self.graphed(rgb, d1)
        g, s_img, s_dep, out = self._graphs[key]
        ops.ingest_raw(bgr, dep, out_rgb=s_img, out_depth=s_dep)
        self.hand._staged_done(staged)
        g.replay()
        return out

    @ops.device_guarded
    def graphed(self, images: torch.Tensor, depth: torch.Tensor):
        """(run, static images, static depth, static LiveOutput): copy new frames into the static inputs and call run()."""
        key = (tuple(images.shape), tuple(depth.shape))
        hit = self._graphs.get(key)
        if hit is None:
            with torch.inference_mode(False), torch.no_grad():
                s_img, s_dep = torch.empty_like(images), torch.empty_like(depth)
                s_img.copy_(images)
                s_dep.copy_(depth)
                n = images.shape[0]
                v0 = self.vertices
                nbytes = (n + 1) * record_bytes(3) + n * v0 * 12
                bufs = (torch.zeros((nbytes,), dtype=torch.uint8, device=self.device),
                        torch.zeros((nbytes,), dtype=torch.uint8, pin_memory=True))
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with ops.launch_cost_hidden():
                    with torch.cuda.stream(side):
                        for _ in range(2):
                            self.forward_device(s_img, s_dep, _buffers=bufs)
                    torch.cuda.current_stream().wait_stream(side)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        out = self.forward_device(s_img, s_dep, _buffers=bufs)
            hit = self._graphs[key] = (g, s_img, s_dep, out)
        g, s_img, s_dep, out = hit
        return g.replay, s_img, s_dep, out


@dataclass
class CropMeshOutput:
    keypoints: torch.Tensor      # [K,21,3] crop (u,v,d) as the network returns it, on the device
    image_uvd: torch.Tensor      # [K,21,3] image (u,v,d) of the (clipped) joints
    xyz_mm: torch.Tensor         # [K,21,3] camera xyz in millimetres
    pose2d: torch.Tensor         # [K,21,2] the lifter's input
    mesh: torch.Tensor           # [K,V,3]: out['mesh'] of a2j_mesh.py:77-80 with perm_reverse, else the lifter's raw [K,V0,3]
    pose3d: torch.Tensor         # [K,21,3]
    raw_mesh: torch.Tensor       # [K,V0,3] the lifter's own output
    host: torch.Tensor           # pinned fp32: keypoints | image_uvd | xyz_mm | mesh | 4 range words (as bits) -- ONE copy; the
    #                              engine's buffer for this batch size: the next call overwrites it (read() returns copies)
    k: int = 0

    def read(self):
        """After the stream is synchronised: (keypoints, image_uvd, xyz_mm, mesh, range words) as fresh CPU tensors."""
        k, j3 = self.k, self.keypoints.shape[1] * 3
        h = self.host
        parts = [h[i * k * j3:(i + 1) * k * j3].reshape(k, -1, 3).clone() for i in range(3)]
        mesh = h[3 * k * j3:-4].reshape(k, -1, 3).clone()
        return parts[0], parts[1], parts[2], mesh, h[-4:].view(torch.int32).tolist()


class CropMeshEngine:
    """The stand-alone mesh demo's loop body (a2j_mesh.py:58-80) as one step on the device: dataset crops -> A2J -> np.clip to
    [0, 176] + convert_joints twice (image uv; camera xyz with the sample's intrinsics -- the dataset's float32 box, fractional
    corners: a2jdataset.py:293) in the aggregation's epilogue -> the lifter's input (predict_mesh, ros_demo.py:148-157) ->
    Pose2Mesh -> the caller's last lines (vertex order, camera offset by the first joint, y / z negated: a2j_mesh.py:77-80) ->
    ONE device -> host copy.  The reference goes to the CPU after A2J, converts in numpy and uploads the normalised joints."""

    def __init__(self, a2j, lifter: Pose2MeshEngine, clamp: bool = True, perm_reverse=None):
        if not _same_device(a2j.device, lifter.device):
            raise ValueError(f"A2J on {a2j.device} but the lifter on {lifter.device}")
        self.a2j, self.lifter, self.device, self.clamp = a2j, lifter, a2j.device, bool(clamp)
        self.perm = None
        if perm_reverse is not None:
            self.perm = torch.as_tensor(perm_reverse).to(torch.int64).to(self.device).contiguous()
            if int(self.perm.max()) >= lifter.graphs[0].v or int(self.perm.min()) < 0:
                raise ValueError("perm_reverse points outside the lifter's finest graph")
        self.vertices = lifter.graphs[0].v if self.perm is None else int(self.perm.shape[0])
        self._block = None
        self._graphs = {}
        self._hosts = {}

    @ops.device_guarded
    def forward_device(self, crops, box_f32, paras, _host=None) -> CropMeshOutput:
        """crops [K,1,176,176] (or [K,4,..] for the RGB-D network), box_f32 [K,4] float32, paras [K,4] float32, on the GPU."""
        k = crops.shape[0]
        if self._block is None:
            self._block = torch.zeros((4,), device=self.device, dtype=torch.int32)
        conv = dict(sample_box=box_f32, sample_paras=paras, clamp_keypoints=self.clamp)
        with ops.range_scope(self._block, on=self.a2j.precision == "f16x3" and self.a2j.note_range):
            kp, img, xyz = self.a2j.forward(crops, convert=conv)
            p2d = ops.joints2d_standardize(img)
            raw, pose3d = self.lifter.forward(p2d)
            mesh = raw if self.perm is None else ops.mesh_finish(raw, self.perm, xyz)
            words = ops.range_check_collect(self._block)
        dev = torch.cat([kp.reshape(-1), img.reshape(-1), xyz.reshape(-1), mesh.reshape(-1), words.view(torch.float32)])
        if _host is None:     # one pinned buffer per batch size, like the live step's: the NEXT eager call with this batch size
            _host = self._hosts.get(dev.numel())      # overwrites it (read() hands out copies)
            if _host is None:
                with torch.inference_mode(False):
                    _host = self._hosts[dev.numel()] = torch.empty((dev.numel(),), dtype=torch.float32, pin_memory=True)
        _host.copy_(dev, non_blocking=True)
        return CropMeshOutput(kp, img, xyz, p2d, mesh, pose3d, raw, _host, k)

    @ops.device_guarded
    def graphed(self, crops, box_f32, paras):
        """(run, static crops, static boxes, static intrinsics, static CropMeshOutput): copy a new batch into the static inputs and
        call run() -- every launch of the step and its copy replay from one hipGraph."""
        key = (tuple(crops.shape),)
        hit = self._graphs.get(key)
        if hit is None:
            with torch.inference_mode(False), torch.no_grad():
                s = [torch.empty_like(t) for t in (crops, box_f32, paras)]
                for a, b in zip(s, (crops, box_f32, paras)):
                    a.copy_(b)
                k = crops.shape[0]
                host = torch.zeros((3 * k * self.a2j.joints * 3 + k * self.vertices * 3 + 4,), dtype=torch.float32, pin_memory=True)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with ops.launch_cost_hidden():
                    with torch.cuda.stream(side):
                        for _ in range(2):
                            self.forward_device(*s, _host=host)
                    torch.cuda.current_stream().wait_stream(side)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        out = self.forward_device(*s, _host=host)
            hit = self._graphs[key] = (g, s[0], s[1], s[2], out)
        g, s_crops, s_box, s_paras, out = hit
        return g.replay, s_crops, s_box, s_paras, out

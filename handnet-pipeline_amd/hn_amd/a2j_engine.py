"""A2J pose network on MI355X: dilated ResNet-50 trunk + 3 conv heads + anchor aggregation.

Mirrors (arithmetically) a2j/a2j.py:194-250 of the reference.  Default precision "f16x3": every convolution with
Cin % 32 == 0 runs hn_conv2d_nhwc_f16x3(_ws / _grouped) on S32 split activations (BatchNorm folded, ReLU /
residual fused in the epilogue, split-K for the 11x11 layers), the stem (Cin 1 or 4) runs the exact f32-MFMA
kernel hn_conv2d_nhwc_f32; precision "f32" keeps everything on the latter.  The regression and depth heads read
the same x4, so their first layers run as ONE 2048 -> 512 conv, and layers 2-4 / the outputs of the three heads
are one grouped launch each.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .forms import engine_forms
from .weights import ConvW, bn_scale_shift, concat_cout, pack_conv, pack_stem_split, strip_prefix

# (planes, blocks, stride of first block, dilation of later blocks)  a2j/resnet.py:109-112
_LAYERS = [(64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 2, 1), (512, 3, 1, 2)]
ANCHORS_PER_CELL = 16


class A2JEngine:
    def __init__(self, state_dict, num_joints: int = 21, rgbd: bool = False, device="cuda", precision="f16x3", forms=None):
        """precision: "f16x3" (split-fp16 operands on the f16 MFMA, fp32-grade results; default)
        or "f32" (exact f32 MFMA).  Layers with Cin % 32 != 0 (the stem) always run in f32."""
        if precision not in ("f32", "f16x3", "f16x1"):
            raise ValueError("precision must be 'f32', 'f16x3' or 'f16x1'")
        # "f16x1": the f16x3 engine with the hi*hi term alone (plain fp16 operands, one MFMA per MAC): the throughput mode
        # SURVEY D6 plans BESIDE the parity mode -- misses the 1e-3 keypoint contract, never a default
        self.terms = 1 if precision == "f16x1" else 3
        precision = "f16x3" if precision == "f16x1" else precision
        self.precision = precision
        sd = strip_prefix(state_dict, "a2j.")
        ops.clear_plan_caches()   # plans are keyed by weight addresses; a rebuilt engine starts clean
        self.device = torch.device(device)
        self.joints = num_joints
        self.rgbd = rgbd
        self.note_range = os.environ.get("HN_CHECK_RANGE", "1") != "0"   # f16x3 range contract (forward_flags)
        fm = engine_forms(forms)   # launch-structure switches (hn_amd/forms.py; never read from the environment here)
        self.group_heads = fm["group_convs"]
        # heterogeneous launches (ops.conv2d_nhwc_multi): the downsample beside conv1 of a block, the classification head
        # beside layer4, the regression / depth towers beside each other (conv_multi = False: the round-3 launch structure)
        self.multi = fm["conv_multi"] and precision == "f16x3"
        p = "Backbone.model."
        dev = self.device

        def cbn(conv, bn, **kw):
            return pack_conv(sd[conv + ".weight"], None, bn_scale_shift(sd, bn), **kw).to(dev)

        # stem: depth replicated to 3 channels in the reference -> fold to one channel
        self.stem = cbn(p + "conv1", p + "bn1", stride=2, pad=3, sum_cin=not rgbd)
        # f16x3 mode: conv1 + bn1 + relu + maxpool as ONE split-precision kernel (the FCOS stem's, hn_conv_stem_pool_f16x3)
        # on the crops' stem image; the f32-MFMA stem + separate pooling pass was 5 % of the batch-64 step
        w1 = sd[p + "conv1.weight"].double()
        self.stem16 = pack_stem_split(w1 if rgbd else w1.sum(dim=1, keepdim=True), bn_scale_shift(sd, p + "bn1")).to(dev) \
            if precision == "f16x3" else None
        self.fuse_stem_pool = fm["fuse_stem_pool"]
        self.blocks = []
        for li, (planes, blocks, stride, dil) in enumerate(_LAYERS, start=1):
            for b in range(blocks):
                q = f"{p}layer{li}.{b}."
                st = stride if b == 0 else 1
                dl = 1 if b == 0 else dil
                blk = {
                    "c1": cbn(q + "conv1", q + "bn1"),
                    "c2": cbn(q + "conv2", q + "bn2", stride=st, pad=dl, dil=dl),
                    "c3": cbn(q + "conv3", q + "bn3"),
                    "ds": cbn(q + "downsample.0", q + "downsample.1", stride=st)
                    if (q + "downsample.0.weight") in sd else None,
                    "tap": li,  # layer index, to tap x3 after layer3
                }
                self.blocks.append(blk)

        def head(name):
            convs = []
            for i in range(1, 5):
                convs.append(pack_conv(sd[f"{name}.conv{i}.weight"], sd[f"{name}.conv{i}.bias"],
                                       bn_scale_shift(sd, f"{name}.bn{i}"), pad=1))
            out = pack_conv(sd[f"{name}.output.weight"], sd[f"{name}.output.bias"], None, pad=1)
            return convs, out

        cls_c, cls_o = head("classificationModel")
        reg_c, reg_o = head("regressionModel")
        dep_c, dep_o = head("DepthRegressionModel")
        self.cls_convs = [c.to(dev) for c in cls_c]
        self.cls_out = cls_o.to(dev)
        self.regdep_conv1 = concat_cout([reg_c[0], dep_c[0]]).to(dev)  # shared-input fusion
        self.reg_convs = [c.to(dev) for c in reg_c[1:]]
        self.dep_convs = [c.to(dev) for c in dep_c[1:]]
        self.reg_out = reg_o.to(dev)
        self.dep_out = dep_o.to(dev)
        if self.cls_out.cout != ANCHORS_PER_CELL * num_joints:
            raise ValueError("checkpoint does not match num_joints")

    # -----------------------------------------------------------------------------------
    def _conv(self, x, cw: ConvW, relu=True, residual=None, tile=0, algo_cin=None, out_f32=False):
        """f16x3 mode: activations travel in the S32 split format (written by each epilogue, read by
        LDS-DMA); only the stem (Cin = 4) runs on the f32 kernel and only head outputs are fp32."""
        s16 = self.precision == "f16x3"
        return ops.conv2d_nhwc(x, cw.w, cw.bias, stride=cw.stride, pad=cw.pad, dil=cw.dil, relu=relu,
                               residual=residual, tile=tile, algo_cin=algo_cin,
                               w16=cw.w16 if s16 else None, out_split=s16 and not out_f32)

    def trunk(self, x, valid=None):
        """x [K,H,W,4] NHWC fp32 -> (x3 [K,H/16,W/16,1024], x4 [K,H/16,W/16,2048]) (S32 in f16x3 mode).  valid: the
        aggregation's per-crop flags; a crop with non-finite pixels is marked 2 in them (NaN keypoints, like the reference)."""
        ops.PROFILE_STAGE = "a2j_trunk"
        if self.stem16 is not None and self.fuse_stem_pool:
            x = ops.conv_stem_pool_split(ops.stem_image_nhwc4(x, valid=valid), self.stem16.w16, self.stem16.bias, 64, r=7, stride=2,
                                         algo_cin=4 if self.rgbd else 3)
        else:
            x = self._conv(x, self.stem, algo_cin=4 if self.rgbd else 3)
            x = ops.maxpool3x3s2_nhwc(x)
        if self.multi and x.shape[0] <= self.MULTI_MAX_CROPS:
            # small batches are launch-bound: the downsample runs beside conv1 and the classification head rides on layer4's
            # launches (measured, A2J alone: 1 crop 0.909 -> 0.799 ms, 2 crops 0.873 -> 0.831, 4 crops even; from 8 crops on
            # the grouped head layers of the other form win -- their members would go split-K one by one here)
            return self._trunk_multi(x)
        x3 = None
        for i, blk in enumerate(self.blocks):
            o = self._conv(x, blk["c1"])
            o = self._conv(o, blk["c2"])
            idn = self._conv(x, blk["ds"], relu=False) if blk["ds"] is not None else x
            x = self._conv(o, blk["c3"], relu=True, residual=idn)
            last_of_layer = (i + 1 == len(self.blocks)) or (self.blocks[i + 1]["tap"] != blk["tap"])
            if blk["tap"] == 3 and last_of_layer:
                x3 = x
        return x3, x

    MULTI_MAX_CROPS = 4

    def _trunk_multi(self, x, ride_cls=True):
        """The trunk as heterogeneous launches: conv1 of a block together with its downsample (both read the block input), and
        -- from the first block of layer4 on, when x3 exists -- one stage of the classification head (four 3x3 convolutions and
        the output convolution, a2j/a2j.py:162-181, which read x3 and nothing of layer4) in each of layer4's launches.  Every
        member computes what its own conv2d_nhwc call would; the library runs members of one tile form in one grid.
        The classification head's output is kept for heads() (self._cls_ready)."""
        cls_stages = [(cw, dict(relu=True)) for cw in self.cls_convs] + [(self.cls_out, dict(relu=False, out_split=False))]

        def launch(items):
            if len(items) > 1:
                return ops.conv2d_nhwc_multi(items)
            x_, cw_, o_ = items[0]
            return [self._conv(x_, cw_, relu=o_.get("relu", False), residual=o_.get("residual"),
                               out_f32=not o_.get("out_split", True))]
        c, stage, x3 = None, 0, None
        for i, blk in enumerate(self.blocks):
            in_l4 = blk["tap"] == 4
            if in_l4 and c is None:
                c = x3 = x                                              # x3: the output of layer3's last block
            items = [(x, blk["c1"], dict(relu=True))]
            if blk["ds"] is not None:
                items.append((x, blk["ds"], dict(relu=False)))

            def ride(items):
                if ride_cls and in_l4 and stage < len(cls_stages):
                    items.append((c, cls_stages[stage][0], cls_stages[stage][1]))
                    return True
                return False
            rides = ride(items)
            outs = launch(items)
            o = outs[0]
            idn = outs[1] if blk["ds"] is not None else x
            if rides:
                c, stage = outs[-1], stage + 1
            items = [(o, blk["c2"], dict(relu=True))]
            rides = ride(items)
            outs = launch(items)
            o = outs[0]
            if rides:
                c, stage = outs[-1], stage + 1
            items = [(o, blk["c3"], dict(relu=True, residual=idn))]
            rides = ride(items)
            outs = launch(items)
            x = outs[0]
            if rides:
                c, stage = outs[-1], stage + 1
        if not ride_cls:
            self._cls_ready = None
            return x3, x
        while stage < len(cls_stages):                                  # (a trunk with a shorter layer4 than the head chain)
            c = launch([(c, cls_stages[stage][0], cls_stages[stage][1])])[0]
            stage += 1
        self._cls_ready = (x3, c)
        return x3, x

    def heads(self, x3, x4):
        ops.PROFILE_STAGE = "a2j_heads"
        ready = getattr(self, "_cls_ready", None)
        self._cls_ready = None
        if self.multi and ready is not None and ready[0] is x3:
            # the classification head has run beside layer4 (_trunk_multi); the regression and depth towers run side by side
            cls = ready[1]
            rd = self._conv(x4, self.regdep_conv1)
            r, d = rd[:, :, :, :8], rd[:, :, :, 8:]
            for cr, cd in zip(self.reg_convs, self.dep_convs):
                r, d = ops.conv2d_nhwc_multi([(r, cr, dict(relu=True)), (d, cd, dict(relu=True))])
            reg, dep = ops.conv2d_nhwc_multi([(r, self.reg_out, dict(relu=False, out_split=False)),
                                              (d, self.dep_out, dict(relu=False, out_split=False))])
            return cls, reg, dep
        c = self._conv(x3, self.cls_convs[0])
        rd = self._conv(x4, self.regdep_conv1)
        # the fused tensor has 512 channels: 0..255 regression tower, 256..511 depth tower
        # (read in place as channel-slice views: no copy)
        if ops.is_split(rd):
            r, d = rd[:, :, :, :8], rd[:, :, :, 8:]
        else:
            r, d = rd[..., :256], rd[..., 256:]
        if self.precision == "f16x3" and self.group_heads:
            # layers 2-4 of the three heads are independent 256->256 3x3 convs on 11x11 maps: ONE launch per
            # layer (layer 2 splits 1 + 2 because r / d are still slices of the fused tensor), outputs 2 + 1
            c = self._conv(c, self.cls_convs[1])
            r, d = ops.conv2d_nhwc_grouped([r, d], [self.reg_convs[0], self.dep_convs[0]], pad=1, relu=True, out_split=True)
            for i in (1, 2):
                c, r, d = ops.conv2d_nhwc_grouped([c, r, d], [self.cls_convs[i + 1], self.reg_convs[i], self.dep_convs[i]],
                                                  pad=1, relu=True, out_split=True)
            cls, dep = ops.conv2d_nhwc_grouped([c, d], [self.cls_out, self.dep_out], pad=1)
            return cls, self._conv(r, self.reg_out, relu=False, out_f32=True), dep
        for cw in self.cls_convs[1:]:
            c = self._conv(c, cw)
        cls = self._conv(c, self.cls_out, relu=False, out_f32=True)
        for cw in self.reg_convs:
            r = self._conv(r, cw)
        for cw in self.dep_convs:
            d = self._conv(d, cw)
        reg = self._conv(r, self.reg_out, relu=False, out_f32=True)
        dep = self._conv(d, self.dep_out, relu=False, out_f32=True)
        return cls, reg, dep

    @ops.device_guarded
    def forward_nhwc(self, x, valid=None, return_heads=False, convert=None):
        """convert (ops.a2j_aggregate's dict: crop_box, paras, clamps): the aggregation's epilogue also writes image (u,v,d)
        and camera xyz in mm -> returns (crop_uvd, image_uvd, xyz_mm or None) instead of crop_uvd alone."""
        with ops.f16_terms(self.terms):
            x3, x4 = self.trunk(x, valid)
            cls, reg, dep = self.heads(x3, x4)
        out = ops.a2j_aggregate(cls, reg, dep, joints=self.joints, stride=16, valid=valid, convert=convert)
        if return_heads:
            if ops.is_split(x3):
                x3, x4 = ops.from_split(x3), ops.from_split(x4)
            return out, (x3, x4), (cls, reg, dep)
        return out

    @ops.device_guarded
    def forward_flags(self, depth, convert=None):
        """The A2J-only entry with the f16x3 range contract: -> (keypoints [K,J,3] on the GPU, flag words [4] int32 on the GPU
        or None in the f32 mode / with HN_CHECK_RANGE=0).  No sync; the drop-in reads the words with the keypoints.
        convert: as forward()."""
        if self.precision != "f16x3" or not self.note_range:
            return self.forward(depth, convert=convert), None
        if getattr(self, "_range_block", None) is None:
            self._range_block = torch.zeros((4,), device=self.device, dtype=torch.int32)
        with ops.range_scope(self._range_block):
            kp = self.forward(depth, convert=convert)
            flags = ops.range_check_collect(self._range_block)
        return kp, flags

    @ops.device_guarded
    def forward(self, depth, valid=None, convert=None):
        """depth [K,1,H,W] (or [K,4,H,W] for RGBD) fp32 on the GPU -> [K,J,3] on the GPU; with convert (forward_nhwc) the
        triple (crop uvd, image uvd, camera xyz in mm or None)."""
        if depth.dim() != 4:
            raise ValueError("expected [K,C,H,W]")
        if not depth.is_cuda:
            raise RuntimeError("A2JEngine needs GPU tensors (no CPU fallback)")
        depth = depth.float().contiguous()
        if self.rgbd:
            x = depth[:, :4].permute(0, 2, 3, 1).contiguous()
        else:
            x = ops.pack_depth_nhwc(depth[:, 0:1].contiguous(), cpad=4)
        if valid is None and self.stem16 is not None:
            # (the A2J-only entry: flags of its own, so that a crop with NaN / inf pixels gives NaN keypoints like
            # a2j/a2j.py:243-250 does)
            valid = torch.ones((depth.shape[0],), device=depth.device, dtype=torch.int32)
        return self.forward_nhwc(x, valid, convert=convert)

    # -----------------------------------------------------------------------------------
    def macs_per_crop(self, h=176, w=176) -> int:
        """Algorithmic MACs exactly as the reference executes them (3-channel stem)."""
        from .ops import conv_out_size

        total = 0
        oh, ow = conv_out_size(h, w, 7, 7, 2, 3, 1)
        total += oh * ow * 64 * 49 * (4 if self.rgbd else 3)
        oh, ow = (oh + 2 - 3) // 2 + 1, (ow + 2 - 3) // 2 + 1
        for blk in self.blocks:
            total += oh * ow * blk["c1"].macs_per_pixel()
            c2 = blk["c2"]
            oh2, ow2 = conv_out_size(oh, ow, 3, 3, c2.stride, c2.pad, c2.dil)
            total += oh2 * ow2 * (c2.macs_per_pixel() + blk["c3"].macs_per_pixel())
            if blk["ds"] is not None:
                total += oh2 * ow2 * blk["ds"].macs_per_pixel()
            oh, ow = oh2, ow2
        px = oh * ow
        for cw in self.cls_convs + [self.cls_out, self.regdep_conv1] + self.reg_convs + self.dep_convs + \
                [self.reg_out, self.dep_out]:
            total += px * cw.macs_per_pixel()
        return total

"""FCOS hand detector on MI355X: ResNet-34-FPN + GroupNorm towers + on-device post-process.

Arithmetic follows fcos_utils/fcos.py:675-767 of the reference (eval branch); default precision "f16x3":
  transform (torchvision GeneralizedRCNNTransform)   -> hn_fcos_preprocess_split (the stem's split fp16 image;
                                                         hn_fcos_preprocess_list for mixed-size batches)
  resnet34 body + FrozenBatchNorm (folded) + FPN      -> hn_conv_stem_f16x3, hn_conv2d_nhwc_f16x3(_ws / _grouped)
                                                         on S32 split activations (+ residual / nearest-2x
                                                         top-down add epilogues)
  4 x (conv3x3 + GroupNorm(32) + ReLU) towers          -> grouped conv launches write the raw fp32 output AND the
                                                         GroupNorm partial sums; hn_groupnorm_finalize_rows32
                                                         builds per-(image, channel) scale/shift tables,
                                                         hn_affine_split_f32 applies them (+ReLU) and emits the
                                                         next conv's S32 input
  cls_logits+hand_lr / bbox_reg+ctrness               -> one Cout=C+2 and one Cout=5 conv
  score/threshold/decode/compaction, batched NMS      -> hn_fcos_candidates, hn_fcos_nms(_ratios)
precision "f32" keeps every convolution on the exact f32-MFMA kernel (hn_conv2d_nhwc_f32; GroupNorm applied on
load by the next conv).  Everything stays on the device; nothing here synchronises with the host.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .forms import engine_forms
from .weights import ConvW, bn_scale_shift, concat_cout, pack_conv, pack_stem_split

IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
SCORE_THRESH = 0.7  # hard-coded in the reference, fcos_utils/fcos.py:600 (ctor args are ignored)
NMS_THRESH = 0.3    # hard-coded, fcos_utils/fcos.py:635
_R34 = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]


def resized_size(h: int, w: int, min_size: int = 800, max_size: int = 1333):
    """Size after GeneralizedRCNNTransform.resize.  torchvision evaluates
    `float / tensor` = tensor.reciprocal() * float in fp32, then floor(size * scale) in fp64."""
    inv_min = torch.tensor(float(min(h, w))).reciprocal()
    inv_max = torch.tensor(float(max(h, w))).reciprocal()
    scale = torch.min(inv_min * float(min_size), inv_max * float(max_size)).item()
    return int(h * scale), int(w * scale)


class FCOSEngine:
    def __init__(self, state_dict, num_classes: int, device="cuda", min_size=800, max_size=1333,
                 precision="f16x3", ext=False, head_streams=None, image_mean=None, image_std=None, forms=None):
        """precision: "f16x3" (split-fp16 operands, fp32-grade; default) or "f32" (exact f32 MFMA).
        image_mean / image_std: the transform's normalisation (fcos.py:501-505; default: ImageNet's)."""
        if precision not in ("f32", "f16x3", "f16x1"):
            raise ValueError("precision must be 'f32', 'f16x3' or 'f16x1'")
        # "f16x1": the f16x3 engine with the hi*hi term alone (plain fp16 operands, one MFMA per MAC): the throughput mode
        # SURVEY D6 plans BESIDE the parity mode -- misses the 1e-3 keypoint contract, never a default
        self.terms = 1 if precision == "f16x1" else 3
        precision = "f16x3" if precision == "f16x1" else precision
        self.precision = precision
        sd = state_dict
        ops.clear_plan_caches()   # plans are keyed by weight addresses; a rebuilt engine starts clean
        dev = torch.device(device)
        self.device = dev
        self.num_classes = num_classes
        self.min_size, self.max_size = min_size, max_size
        self.image_mean = tuple(float(v) for v in (IMAGE_MEAN if image_mean is None else image_mean))
        self.image_std = tuple(float(v) for v in (IMAGE_STD if image_std is None else image_std))
        if len(self.image_mean) != 3 or len(self.image_std) != 3 or any(v == 0.0 for v in self.image_std):
            raise ValueError("image_mean / image_std must hold three values (std non-zero)")
        p = "backbone.body."

        def cbn(conv, bn, **kw):
            return pack_conv(sd[conv + ".weight"], None, bn_scale_shift(sd, bn), **kw).to(dev)

        self.stem = cbn(p + "conv1", p + "bn1", stride=2, pad=3)  # Cin 3 -> padded to 4
        # f16x3 mode: the stem runs on the split kernel too, one filter row (7 taps x 4 channels) per k tile
        self.stem16 = pack_stem_split(sd[p + "conv1.weight"], bn_scale_shift(sd, p + "bn1")).to(dev) \
            if precision == "f16x3" else None
        self.blocks = []
        for li, (planes, blocks, stride) in enumerate(_R34, start=1):
            for b in range(blocks):
                q = f"{p}layer{li}.{b}."
                st = stride if b == 0 else 1
                self.blocks.append({
                    "c1": cbn(q + "conv1", q + "bn1", stride=st, pad=1),
                    "c2": cbn(q + "conv2", q + "bn2", pad=1),
                    "ds": cbn(q + "downsample.0", q + "downsample.1", stride=st)
                    if (q + "downsample.0.weight") in sd else None,
                    "layer": li, "last": b == blocks - 1,
                })
        f = "backbone.fpn."
        self.inner = [pack_conv(sd[f"{f}inner_blocks.{i}.weight"], sd[f"{f}inner_blocks.{i}.bias"]).to(dev)
                      for i in range(3)]
        self.layer = [pack_conv(sd[f"{f}layer_blocks.{i}.weight"], sd[f"{f}layer_blocks.{i}.bias"], pad=1).to(dev)
                      for i in range(3)]
        c, r = "head.classification_head", "head.regression_head"

        def tconv(t, i):
            return pack_conv(sd[f"{t}.conv.{3 * i}.weight"], sd[f"{t}.conv.{3 * i}.bias"], pad=1)

        # layer 0 of both towers reads the same FPN level -> one 256->512 conv, one GN(64 groups)
        self.tower0 = concat_cout([tconv(c, 0), tconv(r, 0)]).to(dev)
        self.cls_tower = [tconv(c, i).to(dev) for i in range(1, 4)]
        self.reg_tower = [tconv(r, i).to(dev) for i in range(1, 4)]
        gn = lambda t, i, k: sd[f"{t}.conv.{3 * i + 1}.{k}"].float()  # noqa: E731
        self.gn0_gamma = torch.cat([gn(c, 0, "weight"), gn(r, 0, "weight")]).to(dev)
        self.gn0_beta = torch.cat([gn(c, 0, "bias"), gn(r, 0, "bias")]).to(dev)
        self.cls_gn = [(gn(c, i, "weight").to(dev), gn(c, i, "bias").to(dev)) for i in range(1, 4)]
        self.reg_gn = [(gn(r, i, "weight").to(dev), gn(r, i, "bias").to(dev)) for i in range(1, 4)]
        self.both_gn = [(torch.cat([gc, gr]), torch.cat([bc, br])) for (gc, bc), (gr, br) in zip(self.cls_gn, self.reg_gn)]
        self.cls_out = concat_cout([
            pack_conv(sd[c + ".cls_logits.weight"], sd[c + ".cls_logits.bias"], pad=1),
            pack_conv(sd[c + ".hand_lr_layer.weight"], sd[c + ".hand_lr_layer.bias"], pad=1)]).to(dev)
        self.reg_out = concat_cout([
            pack_conv(sd[r + ".bbox_reg.weight"], sd[r + ".bbox_reg.bias"], pad=1),
            pack_conv(sd[r + ".bbox_ctrness.weight"], sd[r + ".bbox_ctrness.bias"], pad=1)]).to(dev)
        if self.cls_out.cout != num_classes + 2:
            raise ValueError("checkpoint does not match num_classes")
        # ext=True heads (fcos.py:255-264): dxdy-magnitude (3, ReLU) and contact state (5) from the cls tower
        self.ext = ext
        self.ext_out = concat_cout([
            pack_conv(sd[c + ".hand_dydx_layer.weight"], sd[c + ".hand_dydx_layer.bias"], pad=1),
            pack_conv(sd[c + ".hand_contact_state_layer.weight"], sd[c + ".hand_contact_state_layer.bias"], pad=1),
        ]).to(dev) if ext else None
        self._gn_scratch = {}
        self._side = None
        # launch-structure switches (hn_amd/forms.py: defaults unless a development host changed them; never the environment)
        fm = engine_forms(forms)
        self.head_streams = fm["head_streams"] if head_streams is None else head_streams
        self.group_towers = fm["group_convs"]
        self.fuse_stem_pool = fm["fuse_stem_pool"]   # (results are bit-identical either way)
        self.fuse_last_gn = fm["fuse_last_gn"]       # bit-identical
        self.thin_outputs = fm["thin_outputs"] and self.cls_out.cout <= 16   # (tap form bit-identical, P form to fp32 rounding)
        # conv1 of a downsampling block together with its 1x1 downsample (ops.conv2d_nhwc_multi)
        self.multi = fm["conv_multi_fcos"] and precision == "f16x3"

    # -----------------------------------------------------------------------------------
    def _conv(self, x, cw: ConvW, relu=False, out_f32=False, **kw):
        """f16x3 mode: activations travel in the S32 split format (written by each epilogue, read by
        LDS-DMA); the stem (Cin = 4) runs on the f32 kernel; tower outputs that feed GroupNorm and
        the final head outputs are fp32."""
        s16 = self.precision == "f16x3"
        return ops.conv2d_nhwc(x, cw.w, cw.bias, stride=cw.stride, pad=cw.pad, dil=cw.dil, relu=relu,
                               w16=cw.w16 if s16 else None, out_split=s16 and not out_f32, **kw)

    def geometry(self, h, w):
        oh, ow = resized_size(h, w, self.min_size, self.max_size)
        ph, pw = int(math.ceil(oh / 32) * 32), int(math.ceil(ow / 32) * 32)
        return oh, ow, ph, pw

    def backbone(self, x):
        """x: preprocessed canvas -- fp32 [N,PH,PW,4], or the fp16 stem image [2,N,PH+6,PW+6,4] of
        ops.fcos_preprocess_split (f16x3 mode) -> [P3, P4, P5] (256 channels, strides 8/16/32; S32 in f16x3 mode)."""
        ops.PROFILE_STAGE = "resnet34_body"
        if x.dtype == torch.float16 and self.fuse_stem_pool:
            # conv1 + bn1 + relu + maxpool as one kernel: the 64-channel half-resolution map never reaches HBM
            x = ops.conv_stem_pool_split(x, self.stem16.w16, self.stem16.bias, 64, r=7, stride=2)
        else:
            if x.dtype == torch.float16:
                x = ops.conv_stem_split(x, self.stem16.w16, self.stem16.bias, 64, r=7, stride=2, relu=True)
            else:
                x = self._conv(x, self.stem, relu=True, algo_cin=3)
            x = ops.maxpool3x3s2_nhwc(x)
        feats = []
        for blk in self.blocks:
            if blk["ds"] is not None and self.multi and x.dtype == torch.float16:
                # both read the block input and neither reads the other: one grid (same bits as the two launches)
                o, idn = ops.conv2d_nhwc_multi([(x, blk["c1"], dict(relu=True)), (x, blk["ds"], dict(relu=False))])
            else:
                o = self._conv(x, blk["c1"], relu=True)
                idn = self._conv(x, blk["ds"]) if blk["ds"] is not None else x
            x = self._conv(o, blk["c2"], relu=True, residual=idn)
            if blk["last"] and blk["layer"] >= 2:
                feats.append(x)
        c3, c4, c5 = feats
        ops.PROFILE_STAGE = "fpn"
        lat5 = self._conv(c5, self.inner[2])
        lat4 = self._conv(c4, self.inner[1], residual=lat5, res_upsample=True)
        lat3 = self._conv(c3, self.inner[0], residual=lat4, res_upsample=True)
        if self.precision == "f16x3" and self.group_towers:
            # the three FPN output convs (own weights, own map size) are independent: one grouped launch
            return ops.conv2d_nhwc_grouped([lat3, lat4, lat5], self.layer, pad=1, out_split=True)
        return [self._conv(lat3, self.layer[0]), self._conv(lat4, self.layer[1]), self._conv(lat5, self.layer[2])]

    def _scratch(self, key, need, device):
        """Per-chain GroupNorm scratch (chains may run on different streams)."""
        buf = self._gn_scratch.get(key)
        if buf is None or buf.numel() < need:
            buf = self._gn_scratch[key] = torch.empty((need,), device=device, dtype=torch.float32)
        return buf

    def _gn(self, x, gamma, beta, groups, key=0):
        n, h, w, c = x.shape
        need = ops._lib.load().hn_groupnorm_scratch_floats(n, h * w, c, groups)
        return ops.groupnorm_affine(x, gamma, beta, groups=groups, scratch=self._scratch(key, need, x.device))

    # -----------------------------------------------------------------------------------
    # head: conv -> GroupNorm -> ReLU -> conv.  Each tower conv writes its raw fp32 output, GroupNorm becomes
    # a per-(image, channel) affine, and
    #   f16x3: the conv epilogue also emits the GroupNorm partial sums (no second read of its output), a
    #          tiny finalize kernel builds the tables, and one hn_affine_split_f32 pass applies affine + ReLU
    #          and emits the S32 input of the next conv (whose hot loop is then pure DMA + MFMA);
    #   f32  : a statistics pass reads the output back; the next conv applies the affine + ReLU while
    #          staging its input (in_affine).
    # -----------------------------------------------------------------------------------
    def _conv_gn(self, a, cw, gamma, beta, groups, n, hw, key):
        """tower conv + the GroupNorm tables of its output -> (raw output, scale, shift)"""
        s16 = self.precision == "f16x3"
        if s16 and hw >= 32:
            part = self._scratch(key, ops.gn_rows32_scratch_floats(n * hw, cw.cout), cw.w.device)
            y = self._conv(a, cw, out_f32=True, gn_partial=part)
            return (y, *ops.groupnorm_finalize_rows32(part, gamma, beta, n, hw, groups))
        y = self._conv(a, cw, out_f32=True) if s16 else self._conv(a[0], cw, in_scale=a[1], in_shift=a[2])
        return (y, *self._gn(y, gamma, beta, groups, key))

    def _act(self, x, scale, shift):
        """GroupNorm affine + ReLU, materialised (S32) or deferred to the next conv's load (f32 mode)."""
        return ops.to_split(x, scale, shift, relu=True) if self.precision == "f16x3" else (x, scale, shift)

    def _out_conv(self, a, cw, **kw):
        prev, ops.PROFILE_STAGE = ops.PROFILE_STAGE, "head_outputs"
        try:
            return self._out_conv_impl(a, cw, **kw)
        finally:
            ops.PROFILE_STAGE = prev

    def _out_conv_impl(self, a, cw, **kw):
        if self.precision == "f16x3":
            return self._conv(a, cw, out_f32=True, **kw)
        return self._conv(a[0], cw, in_scale=a[1], in_shift=a[2], **kw)

    def _tower0(self, feat, key):
        """Layer 0 of BOTH towers on one FPN level: one 256->512 conv, one GroupNorm with 64 groups."""
        n, fh, fw = feat.shape[:3]
        if self.precision == "f16x3":
            return self._conv_gn(feat, self.tower0, self.gn0_gamma, self.gn0_beta, 64, n, fh * fw, key)
        t0 = self._conv(feat, self.tower0, out_f32=True)
        return (t0, *self._gn(t0, self.gn0_gamma, self.gn0_beta, 64, key))

    def _cls_chain(self, t0, sc, sh, key):
        n, fh, fw = t0.shape[:3]
        x, sc, sh = t0[..., :256], sc[:, :256], sh[:, :256]
        for cw, (g, b) in zip(self.cls_tower, self.cls_gn):
            x, sc, sh = self._conv_gn(self._act(x, sc, sh), cw, g, b, 32, n, fh * fw, key)
        a = self._act(x, sc, sh)
        cls_lr = self._out_conv(a, self.cls_out)
        ext = self._out_conv(a, self.ext_out, relu_cols=3) if self.ext else None
        return cls_lr, ext

    def _reg_chain(self, t0, sc, sh, key):
        n, fh, fw = t0.shape[:3]
        x, sc, sh = t0[..., 256:], sc[:, 256:], sh[:, 256:]
        for cw, (g, b) in zip(self.reg_tower, self.reg_gn):
            x, sc, sh = self._conv_gn(self._act(x, sc, sh), cw, g, b, 32, n, fh * fw, key)
        return self._out_conv(self._act(x, sc, sh), self.reg_out, relu_cols=4)

    def head_level(self, feat, key=0):
        """One FPN level -> (cls_lr [N,h,w,C+2], reg_ctr [N,h,w,5], ext [N,h,w,8] or None) raw fp32 conv outputs."""
        t0, sc, sh = self._tower0(feat, key)
        n, fh, fw = t0.shape[:3]
        hw = fh * fw
        if self.precision != "f16x3" or hw < 32 or not self.group_towers:
            cls_lr, ext = self._cls_chain(t0, sc, sh, key)
            return cls_lr, self._reg_chain(t0, sc, sh, key), ext
        # f16x3: layer k of the cls tower and of the reg tower are independent and identical in shape -> ONE
        # launch each (gridDim.z = 2) whose two outputs stay stacked as [N,h,w,512] like tower0's, so that
        # ONE GroupNorm finalize (64 groups) and ONE affine+split pass serve both towers of a layer.
        part = self._scratch((key, "g2"), ops.gn_rows32_scratch_floats(n * hw, 512), t0.device)
        for cwc, cwr, (g2, b2) in zip(self.cls_tower, self.reg_tower, self.both_gn):
            a = self._act(t0, sc, sh)                                   # S32 [N,h,w,16,2,32]
            t0 = torch.empty((n, fh, fw, 512), device=t0.device, dtype=torch.float32)
            ops.conv2d_nhwc_grouped([a[:, :, :, :8], a[:, :, :, 8:]], [cwc, cwr], pad=1, outs=[t0, t0],
                                    out_channel_offsets=[0, 256], gn=[(part, 0), (part, 32)], gn_units=64)
            sc, sh = ops.groupnorm_finalize_rows32(part, g2, b2, n, hw, 64)
        a = self._act(t0, sc, sh)
        ac, ar = a[:, :, :, :8], a[:, :, :, 8:]
        cls_lr = self._out_conv(ac, self.cls_out)
        ext = self._out_conv(ac, self.ext_out, relu_cols=3) if self.ext else None
        reg_ctr = self._out_conv(ar, self.reg_out, relu_cols=4)
        return cls_lr, reg_ctr, ext

    def heads_grouped(self, feats):
        """All FPN levels in lockstep (f16x3): the tower weights are shared across levels and the cls / reg towers are
        independent, so layer k of both towers on all levels is ONE grouped launch (gridDim.z = 2 x levels, members of
        different map sizes); tower0 and each output conv are one launch over the levels.  6 conv launches per frame
        batch instead of 18 -- and grids that fill the chip at batch 1."""
        L = len(feats)
        n = feats[0].shape[0]
        dev = feats[0].device
        dims = [tuple(f.shape[1:3]) for f in feats]
        hw = [h * w for h, w in dims]
        parts = [self._scratch(("lv", l), ops.gn_rows32_scratch_floats(n * hw[l], 512), dev) for l in range(L)]
        new_t = lambda: [torch.empty((n, h, w, 512), device=dev, dtype=torch.float32) for h, w in dims]  # noqa: E731
        t = new_t()
        ops.PROFILE_STAGE = "towers"
        ops.conv2d_nhwc_grouped(feats, [self.tower0] * L, pad=1, outs=t, gn=[(parts[l], 0) for l in range(L)], gn_units=64)
        # GroupNorm finalize and the affine + ReLU + split pass: ONE launch each over the levels (the library falls back
        # to per-level launches of the apply pass for tensors beyond the caches, where streaming stores win)
        aff = ops.groupnorm_finalize_rows32_levels(parts, self.gn0_gamma, self.gn0_beta, n, hw, 64)
        for cwc, cwr, (g2, b2) in zip(self.cls_tower, self.reg_tower, self.both_gn):
            a = ops.to_split_levels(t, aff, relu=True)                  # S32 [N,h,w,16,2,32]: cls blocks 0-7, reg 8-15
            t = new_t()
            ops.conv2d_nhwc_grouped([x[:, :, :, :8] for x in a] + [x[:, :, :, 8:] for x in a], [cwc] * L + [cwr] * L,
                                    pad=1, outs=t + t, out_channel_offsets=[0] * L + [256] * L,
                                    gn=[(parts[l], 0) for l in range(L)] + [(parts[l], 32) for l in range(L)], gn_units=64)
            aff = ops.groupnorm_finalize_rows32_levels(parts, g2, b2, n, hw, 64)
        ops.PROFILE_STAGE = "head_outputs"
        if (self.thin_outputs and self.fuse_last_gn and self.terms == 3 and not self.ext
                and ops.thin_affine_applies(t, self.cls_out) and ops.thin_affine_applies(t, self.reg_out)):
            # the last GroupNorm apply pass happens on the head-output kernels' fragments (no 8 bytes per element round trip)
            cls_lr = ops.conv3x3_thin_affine_levels(t, aff, 0, self.cls_out)
            reg_ctr = ops.conv3x3_thin_affine_levels(t, aff, 256, self.reg_out, relu_cols=4)
            return list(zip(cls_lr, reg_ctr, [None] * L))
        ops.PROFILE_STAGE = "towers"
        a = ops.to_split_levels(t, aff, relu=True)
        ac, ar = [x[:, :, :, :8] for x in a], [x[:, :, :, 8:] for x in a]
        ops.PROFILE_STAGE = "head_outputs"
        if self.thin_outputs and self.terms == 3:   # <= 16 output channels: the thin-N kernels (csrc/conv3x3_thin.hip)
            # one launch for the two (ext: three) filter banks where they run the tap kernel (a single frame); else one each
            members = [(ac, self.cls_out, 0)] + ([(ac, self.ext_out, 3)] if self.ext else []) + [(ar, self.reg_out, 4)]
            outs = ops.conv3x3_thin_levels_group(members)
            cls_lr, reg_ctr = outs[0], outs[-1]
            ext = outs[1] if self.ext else [None] * L
        else:
            cls_lr = ops.conv2d_nhwc_grouped(ac, [self.cls_out] * L, pad=1)
            ext = ops.conv2d_nhwc_grouped(ac, [self.ext_out] * L, pad=1, relu_cols=3) if self.ext else [None] * L
            reg_ctr = ops.conv2d_nhwc_grouped(ar, [self.reg_out] * L, pad=1, relu_cols=4)
        return list(zip(cls_lr, reg_ctr, ext))

    def heads(self, feats):
        """All levels.  With head_streams > 1 the 2 x levels independent tower chains are spread over side
        streams so that the tail of one convolution's grid (e.g. 1700 workgroups on 512 slots at the
        stride-16 level) is filled by another chain's workgroups."""
        ops.PROFILE_STAGE = "towers"
        if (self.head_streams <= 1 and self.group_towers and self.precision == "f16x3" and 2 * len(feats) <= 6
                and all(f.shape[1] * f.shape[2] >= 32 for f in feats)):
            return self.heads_grouped(feats)
        if self.head_streams <= 1:
            return [self.head_level(f, key=i) for i, f in enumerate(feats)]
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.head_streams)]
        lv = [self._tower0(f, ("t0", i)) for i, f in enumerate(feats)]
        fork = torch.cuda.Event()
        fork.record(main)
        chains = [(i, kind) for i in range(len(feats)) for kind in ("cls", "reg")]
        results = {}
        for ci, (i, kind) in enumerate(chains):
            st = self._side[ci % len(self._side)]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                if kind == "cls":
                    results[(i, "cls")] = self._cls_chain(*lv[i], key=("cls", i))
                else:
                    results[(i, "reg")] = self._reg_chain(*lv[i], key=("reg", i))
        for st in self._side:
            main.wait_stream(st)
        outs = []
        for i in range(len(feats)):
            cls_lr, ext = results[(i, "cls")]
            reg_ctr = results[(i, "reg")]
            for t in (cls_lr, ext, reg_ctr):
                if t is not None:
                    t.record_stream(main)  # allocated on a side stream, consumed on the main one
            outs.append((cls_lr, reg_ctr, ext))
        return outs

    def list_geometry(self, images):
        """torchvision batch_images for a list of [3,h_i,w_i] images: per-image (h, w, oh, ow) and the common
        canvas (max resized size rounded up to 32), fcos_utils/fcos.py:702-709."""
        geom = []
        for img in images:
            if img.dim() != 3 or img.shape[0] != 3:
                raise ValueError("expected a list of [3,H,W] images")
            h, w = int(img.shape[1]), int(img.shape[2])
            geom.append((h, w) + resized_size(h, w, self.min_size, self.max_size))
        ph = int(math.ceil(max(g[2] for g in geom) / 32) * 32)
        pw = int(math.ceil(max(g[3] for g in geom) / 32) * 32)
        return geom, ph, pw

    @ops.device_guarded
    def forward_heads(self, images):
        with ops.f16_terms(self.terms):
            return self._forward_heads(images)

    def _forward_heads(self, images):
        """images [N,3,H,W] fp32 0..1 on the GPU (or a list of [3,h_i,w_i] images of different sizes)
        -> per-level head tensors + geometry."""
        if not torch.is_tensor(images):
            geom, ph, pw = self.list_geometry(images)
            x = ops.fcos_preprocess_list(images, geom, ph, pw, self.image_mean, self.image_std, split=self.precision == "f16x3")
            oh, ow = [g[2] for g in geom], [g[3] for g in geom]
        else:
            if images.dim() != 4 or images.shape[1] != 3:
                raise ValueError("expected [N,3,H,W]")
            n, _, h, w = images.shape
            oh, ow, ph, pw = self.geometry(h, w)
            pre = ops.fcos_preprocess_split if self.precision == "f16x3" else ops.fcos_preprocess
            x = pre(images.float().contiguous(), oh, ow, ph, pw, self.image_mean, self.image_std)
        feats = self.backbone(x)
        outs = self.heads(feats)
        strides = [ph // f.shape[1] for f in feats]
        self._ext_levels = [o[2] for o in outs] if self.ext else None
        return [o[0] for o in outs], [o[1] for o in outs], strides, (oh, ow, ph, pw)

    @ops.device_guarded
    def detect(self, images, cand=None, det=None, nms_scratch=None):
        """Full detector on the device -> ops.Detections (fixed capacity, score-ordered).  images: [N,3,H,W], or a
        list of [3,h_i,w_i] tensors of different sizes (each resized on its own, boxes rescaled per image)."""
        cls_lr, reg_ctr, strides, (oh, ow, ph, pw) = self.forward_heads(images)
        cand = ops.fcos_candidates(cls_lr, reg_ctr, strides, self.num_classes, SCORE_THRESH, out=cand)
        if det is None:   # (rows at or beyond count[i] stay undefined: nobody downstream reads them)
            det = ops.alloc_detections(cand.scores.shape[0], cand.scores.shape[1], cand.scores.device, zero=False)
        # resize_boxes (fcos.py:770-783): fp32 tensor / fp32 tensor
        if not torch.is_tensor(images):
            hs = torch.tensor([float(i.shape[1]) for i in images]) / torch.tensor([float(v) for v in oh])
            ws = torch.tensor([float(i.shape[2]) for i in images]) / torch.tensor([float(v) for v in ow])
            ratios = torch.stack([hs, ws], dim=1).contiguous().to(self.device)
            return ops.fcos_nms(cand, NMS_THRESH, 1.0, 1.0, scratch=nms_scratch, out=det, ratios=ratios), cand
        n, _, h, w = images.shape
        ratio_h = (torch.tensor(float(h)) / torch.tensor(float(oh))).item()
        ratio_w = (torch.tensor(float(w)) / torch.tensor(float(ow))).item()
        det = ops.fcos_nms(cand, NMS_THRESH, ratio_h, ratio_w, scratch=nms_scratch, out=det)
        return det, cand

    @ops.device_guarded
    def detect_ext(self, images):
        """ext=True detector: (Detections, Candidates, contacts [N,cap] int32, dxdymags [N,cap,3])."""
        if not self.ext:
            raise RuntimeError("engine was built without the ext heads")
        det, cand = self.detect(images)
        contacts, dxdymags = ops.fcos_ext_gather(self._ext_levels, det, cand)
        return det, cand, contacts, dxdymags

    # -----------------------------------------------------------------------------------
    def macs_per_frame(self, h=480, w=640) -> int:
        """Algorithmic conv MACs exactly as the reference executes them (3-channel stem, padded canvas)."""
        _, _, ph, pw = self.geometry(h, w)
        total = 0
        sh, sw = ph // 2, pw // 2
        total += sh * sw * 64 * 49 * 3
        sh, sw = sh // 2, sw // 2
        sizes = {}
        for blk in self.blocks:
            st = blk["c1"].stride
            sh, sw = sh // st, sw // st
            total += sh * sw * (blk["c1"].macs_per_pixel() + blk["c2"].macs_per_pixel())
            if blk["ds"] is not None:
                total += sh * sw * blk["ds"].macs_per_pixel()
            sizes[blk["layer"]] = (sh, sw)
        pts = 0
        for i, li in enumerate((2, 3, 4)):
            a, b = sizes[li]
            total += a * b * (self.inner[i].macs_per_pixel() + self.layer[i].macs_per_pixel())
            pts += a * b
        per_pt = self.tower0.macs_per_pixel() + sum(c.macs_per_pixel() for c in self.cls_tower + self.reg_tower)
        per_pt += self.cls_out.macs_per_pixel() + self.reg_out.macs_per_pixel()
        per_pt += self.ext_out.macs_per_pixel() if self.ext else 0
        return total + pts * per_pt

"""Oracle (CPU, fp32, plain torch ops) for the FCOS hand detector.  Test infrastructure only.

In-repo pieces restated from the reference source (pinned by tests/golden/fcos_*.npz, which
were produced by running the reference's own classes):
  heads          fcos_utils/fcos.py:267-329 (classification), :373-395 (regression)
  anchors        fcos_utils/anchor_utils.py:56-72, 82-132
  box decode     fcos_utils/det_utils.py:266-294
  post-process   fcos_utils/fcos.py:572-659, resize_boxes :770-783, psum :786-790
  wiring         fcos_utils/fcos.py:675-767 (eval branch)
torchvision-0.11.3 pieces the reference only calls (PARITY UNPINNED -- torchvision is not
vendored, not installed, and the reference has no fixtures for them; restated from the
published algorithm and anchored on the call sites fcos.py:476,505,635,709,737):
  GeneralizedRCNNTransform(800, 1333), resnet_fpn_backbone('resnet34', returned_layers=
  [2,3,4]) with FrozenBatchNorm2d, FeaturePyramidNetwork + LastLevelMaxPool, batched_nms.
"""
from __future__ import annotations

import ctypes
import math
import subprocess
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
IMAGE_MEAN = [0.485, 0.456, 0.406]
IMAGE_STD = [0.229, 0.224, 0.225]
SCORE_THRESH = 0.7   # hard-coded, fcos_utils/fcos.py:600
NMS_THRESH = 0.3     # hard-coded, fcos_utils/fcos.py:635


# ---------------------------------------------------------------------------------------
# torchvision GeneralizedRCNNTransform (min_size 800, max_size 1333, size_divisible 32)
# ---------------------------------------------------------------------------------------
def transform(images, min_size=800, max_size=1333, size_divisible=32, image_mean=None, image_std=None):
    """list of [3,H,W] in 0..1 -> (tensor [N,3,PH,PW], image_sizes [(h,w)]).  image_mean / image_std: the FCOS ctor's
    (fcos.py:501-505), default ImageNet's."""
    mean = torch.tensor(IMAGE_MEAN if image_mean is None else list(image_mean))[:, None, None]
    std = torch.tensor(IMAGE_STD if image_std is None else list(image_std))[:, None, None]
    out, sizes = [], []
    for img in images:
        img = (img - mean) / std
        h, w = img.shape[-2:]
        im_shape = torch.tensor([h, w])
        mn = torch.min(im_shape).to(dtype=torch.float32)
        mx = torch.max(im_shape).to(dtype=torch.float32)
        scale = torch.min(min_size / mn, max_size / mx).item()
        img = F.interpolate(img[None], size=None, scale_factor=scale, mode="bilinear",
                            recompute_scale_factor=True, align_corners=False)[0]
        out.append(img)
        sizes.append((img.shape[-2], img.shape[-1]))
    mh = max(s[0] for s in sizes)
    mw = max(s[1] for s in sizes)
    ph = int(math.ceil(mh / size_divisible) * size_divisible)
    pw = int(math.ceil(mw / size_divisible) * size_divisible)
    batched = torch.zeros((len(out), 3, ph, pw), dtype=out[0].dtype)   # (fp64 when the caller passes fp64 images)
    for i, img in enumerate(out):
        batched[i, :, : img.shape[1], : img.shape[2]].copy_(img)
    return batched, sizes


def resized_size(h, w, min_size=800, max_size=1333):
    """Output (oh, ow) of the transform's resize for an h x w image, without running it."""
    # python-number / tensor == tensor.reciprocal() * number in torch: keep that exact form
    scale = torch.min(min_size / torch.tensor(float(min(h, w))), max_size / torch.tensor(float(max(h, w)))).item()
    return int(h * scale), int(w * scale)


# ---------------------------------------------------------------------------------------
# ResNet-34 body + FrozenBatchNorm2d + FPN
# ---------------------------------------------------------------------------------------
def _frozen_bn(x, sd, name, eps=1e-5):
    w = sd[name + ".weight"].reshape(1, -1, 1, 1)
    b = sd[name + ".bias"].reshape(1, -1, 1, 1)
    rv = sd[name + ".running_var"].reshape(1, -1, 1, 1)
    rm = sd[name + ".running_mean"].reshape(1, -1, 1, 1)
    scale = w * (rv + eps).rsqrt()
    bias = b - rm * scale
    return x * scale + bias


def _basic_block(x, sd, p, stride):
    out = F.relu(_frozen_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1), sd, p + "bn1"))
    out = _frozen_bn(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2")
    if (p + "downsample.0.weight") in sd:
        identity = _frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1")
    else:
        identity = x
    return F.relu(out + identity)


_R34 = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]


def body(x, sd, p="backbone.body."):
    """ResNet-34 trunk -> [C2, C3, C4, C5] (strides 4, 8, 16, 32).  Stem and layer1-3 are cross-checked against the
    reference's in-tree a2j/resnet.py ResNet(BasicBlock) (tests/golden/resnet34_intree.npz); layer4's stride-2
    first block is torchvision's (the in-tree class uses stride 1 + dilation there)."""
    x = F.relu(_frozen_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3), sd, p + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    cs = []
    for li, (planes, blocks, stride) in enumerate(_R34, start=1):
        for b in range(blocks):
            x = _basic_block(x, sd, f"{p}layer{li}.{b}.", stride if b == 0 else 1)
        cs.append(x)
    return cs


def fpn(cs, sd):
    """torchvision-0.11.3 ops/feature_pyramid_network.py FeaturePyramidNetwork.forward on [C3, C4, C5] (extra block
    unused here: the reference builds resnet_fpn_backbone(returned_layers=[2,3,4]) and reads levels '0','1','2'):
    lateral 1x1, top-down `lat + interpolate(last, size=lat.shape[-2:], mode='nearest')` from the coarsest level,
    3x3 output conv per level."""
    f = "backbone.fpn."

    def inner(i, t):
        return F.conv2d(t, sd[f"{f}inner_blocks.{i}.weight"], sd[f"{f}inner_blocks.{i}.bias"])

    def layer(i, t):
        return F.conv2d(t, sd[f"{f}layer_blocks.{i}.weight"], sd[f"{f}layer_blocks.{i}.bias"], padding=1)

    last = inner(2, cs[2])
    results = [layer(2, last)]
    for idx in (1, 0):
        lat = inner(idx, cs[idx])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        results.insert(0, layer(idx, last))
    return results


def backbone(x, sd):
    """[N,3,PH,PW] -> OrderedDict('0','1','2','pool') of 256-channel maps (strides 8,16,32,64)."""
    results = fpn(body(x, sd)[1:], sd)
    results.append(F.max_pool2d(results[-1], 1, 2, 0))
    return OrderedDict(zip(["0", "1", "2", "pool"], results))


# ---------------------------------------------------------------------------------------
# heads (fcos_utils/fcos.py:267-329, 373-395); ext=True adds the contact-state / dxdy outputs (:299-320)
# ---------------------------------------------------------------------------------------
def _tower(x, sd, name):
    for i in range(4):
        x = F.conv2d(x, sd[f"{name}.conv.{3 * i}.weight"], sd[f"{name}.conv.{3 * i}.bias"], padding=1)
        x = F.relu(F.group_norm(x, 32, sd[f"{name}.conv.{3 * i + 1}.weight"], sd[f"{name}.conv.{3 * i + 1}.bias"], 1e-5))
    return x


def _flatten(t, k):
    n, _, h, w = t.shape
    return t.view(n, -1, k, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, k)


def head(features, sd, num_classes, ext=False):
    cls_all, lr_all, reg_all, ctr_all, contact_all, dxdy_all = [], [], [], [], [], []
    c = "head.classification_head"
    r = "head.regression_head"
    for feat in features:
        ct = _tower(feat, sd, c)
        cls_all.append(_flatten(F.conv2d(ct, sd[c + ".cls_logits.weight"], sd[c + ".cls_logits.bias"], padding=1), num_classes))
        lr_all.append(_flatten(F.conv2d(ct, sd[c + ".hand_lr_layer.weight"], sd[c + ".hand_lr_layer.bias"], padding=1), 2))
        if ext:
            d = F.relu(F.conv2d(ct, sd[c + ".hand_dydx_layer.weight"], sd[c + ".hand_dydx_layer.bias"], padding=1))
            d = torch.cat([d[:, 0].unsqueeze(1), 0.1 * F.normalize(d[:, 1:], p=2, dim=1)], dim=1)  # fcos.py:301-303
            dxdy_all.append(_flatten(d, 3))
            contact_all.append(_flatten(F.conv2d(ct, sd[c + ".hand_contact_state_layer.weight"],
                                                 sd[c + ".hand_contact_state_layer.bias"], padding=1), 5))
        rt = _tower(feat, sd, r)
        reg_all.append(_flatten(F.relu(F.conv2d(rt, sd[r + ".bbox_reg.weight"], sd[r + ".bbox_reg.bias"], padding=1)), 4))
        ctr_all.append(_flatten(F.conv2d(rt, sd[r + ".bbox_ctrness.weight"], sd[r + ".bbox_ctrness.bias"], padding=1), 1))
    out = {"cls_logits": torch.cat(cls_all, 1), "hand_lr": torch.cat(lr_all, 1),
           "bbox_regression": torch.cat(reg_all, 1), "bbox_ctrness": torch.cat(ctr_all, 1)}
    if ext:
        out["hand_contact_state"] = torch.cat(contact_all, 1)
        out["hand_dxdy"] = torch.cat(dxdy_all, 1)
    return out


# ---------------------------------------------------------------------------------------
# anchors / decode / post-process
# ---------------------------------------------------------------------------------------
def anchors_for(image_size, grid_sizes, sizes=(8, 16, 32)):
    """fcos_utils/anchor_utils.py:56-132 with sizes ((8,),(16,),(32,)), aspect ratio 1."""
    out = []
    for (gh, gw), size in zip(grid_sizes, sizes):
        sh, sw = image_size[0] // gh, image_size[1] // gw
        scales = torch.as_tensor([size], dtype=torch.float32)
        ar = torch.as_tensor([1.0], dtype=torch.float32)
        h_r = torch.sqrt(ar)
        w_r = 1 / h_r
        ws = (w_r[:, None] * scales[None, :]).view(-1)
        hs = (h_r[:, None] * scales[None, :]).view(-1)
        base = (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()
        sx = torch.arange(0, gw, dtype=torch.int32) * sw
        sy = torch.arange(0, gh, dtype=torch.int32) * sh
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
    return torch.cat(out)


def decode_single(rel_codes, boxes):
    """fcos_utils/det_utils.py:266-294, normalize_by_size=True."""
    boxes = boxes.to(rel_codes.dtype)
    ctr_x = 0.5 * (boxes[:, 0] + boxes[:, 2])
    ctr_y = 0.5 * (boxes[:, 1] + boxes[:, 3])
    bw = boxes[:, 2] - boxes[:, 0]
    bh = boxes[:, 3] - boxes[:, 1]
    rel = rel_codes * torch.stack((bw, bh, bw, bh), dim=1)
    return torch.stack((ctr_x - rel[:, 0], ctr_y - rel[:, 1], ctr_x + rel[:, 2], ctr_y + rel[:, 3]), dim=1)


_nms_lib = None


def _load_nms():
    global _nms_lib
    if _nms_lib is None:
        so = HERE / "_build" / "liboracle_nms.so"
        if not so.exists():
            subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
        lib = ctypes.CDLL(str(so))
        lib.oracle_nms_f32.restype = ctypes.c_int64
        lib.oracle_nms_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p]
        _nms_lib = lib
    return _nms_lib


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms (CPU kernel) -> int64 indices, descending score."""
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    keep = np.zeros((max(1, len(s)),), dtype=np.int64)
    k = _load_nms().oracle_nms_f32(b.ctypes.data, s.ctypes.data, len(s), float(iou_threshold), keep.ctypes.data)
    return torch.from_numpy(keep[:k].copy())


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision-0.11.3 ops.boxes.batched_nms (vanilla above 4000 elements, else coordinate trick)."""
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for class_id in torch.unique(idxs):
            curr = torch.where(idxs == class_id)[0]
            ck = nms(boxes[curr], scores[curr], iou_threshold)
            keep_mask[curr[ck]] = True
        keep = torch.where(keep_mask)[0]
        return keep[scores[keep].sort(descending=True, stable=True)[1]]
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def resize_boxes(boxes, original_size, new_size):
    """fcos_utils/fcos.py:770-783."""
    rh = torch.tensor(new_size[0], dtype=torch.float32) / torch.tensor(original_size[0], dtype=torch.float32)
    rw = torch.tensor(new_size[1], dtype=torch.float32) / torch.tensor(original_size[1], dtype=torch.float32)
    x0, y0, x1, y1 = boxes.unbind(1)
    return torch.stack((x0 * rw, y0 * rh, x1 * rw, y1 * rh), dim=1)


def candidates(head_out, anchors, num_anchors_per_level):
    """fcos_utils/fcos.py:591-628 for every image: boxes/scores/labels/sides/level that pass 0.7."""
    cls, reg, ctr, lr = (head_out[k] for k in ("cls_logits", "bbox_regression", "bbox_ctrness", "hand_lr"))
    scores = torch.sqrt(torch.sigmoid(cls) * torch.sigmoid(ctr))
    smax, lmax = torch.max(scores, dim=-1)
    masks = smax > SCORE_THRESH
    _, sides = torch.max(torch.sigmoid(lr), dim=-1)
    level = torch.zeros(anchors.shape[0])
    starts = np.cumsum([0] + list(num_anchors_per_level))
    for i in range(1, len(starts) - 1):
        level[starts[i]: starts[i + 1]] = i
    out = []
    for n in range(cls.shape[0]):
        m = masks[n]
        out.append({"boxes": decode_single(reg[n], anchors)[m], "scores": smax[n][m], "labels": lmax[n][m],
                    "sides": sides[n][m], "feature_idx": level[m], "index": torch.where(m)[0]})
        if "hand_contact_state" in head_out:  # fcos.py:605-607,631-633
            _, contact = torch.max(torch.sigmoid(head_out["hand_contact_state"][n]), dim=-1)
            out[-1]["contacts"] = contact[m]
            out[-1]["dxdymags"] = head_out["hand_dxdy"][n][m]
    return out


def postprocess(cands, image_sizes, original_sizes):
    """NMS (fcos.py:635), gather (:649-657), rescale (:661-669)."""
    dets = []
    for c, im_s, o_s in zip(cands, image_sizes, original_sizes):
        keep = batched_nms(c["boxes"], c["scores"], c["labels"], NMS_THRESH)
        dets.append({"boxes": resize_boxes(c["boxes"][keep], im_s, o_s), "scores": c["scores"][keep],
                     "labels": c["labels"][keep], "sides": c["sides"][keep].reshape(-1),
                     "feature_idx": c["feature_idx"][keep].reshape(-1), "keep": keep})
        if "contacts" in c:  # ext=True dict (fcos.py:637-647)
            dets[-1]["contacts"] = c["contacts"][keep].reshape(-1)
            dets[-1]["dxdymags"] = c["dxdymags"][keep]
    return dets


def fcos_forward(images, sd, num_classes=3, return_intermediates=False, ext=False, image_mean=None, image_std=None):
    """fcos_utils/fcos.py:675-767 (eval): list of [3,H,W] -> list of detection dicts."""
    with torch.no_grad():
        original_sizes = [tuple(img.shape[-2:]) for img in images]
        x, image_sizes = transform(images, image_mean=image_mean, image_std=image_std)
        feats = list(backbone(x, sd).values())[:-1]
        ho = head(feats, sd, num_classes, ext)
        grid = [tuple(f.shape[-2:]) for f in feats]
        anchors = anchors_for(tuple(x.shape[-2:]), grid)
        cands = candidates(ho, anchors, [g[0] * g[1] for g in grid])
        dets = postprocess(cands, image_sizes, original_sizes)
    if return_intermediates:
        return dets, {"x": x, "features": feats, "head": ho, "anchors": anchors, "candidates": cands,
                      "image_sizes": image_sizes}
    return dets

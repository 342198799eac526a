"""Oracle (CPU, fp32, plain torch ops) for the A2J pose network.  Test infrastructure only.

Follows the reference file by file:
  trunk       a2j/a2j.py:194-210  (ResNetBackBone.forward) over
              a2j/resnet.py:61-96 (Bottleneck), :101-131 (ResNet ctor: layer4 stride 1,
              dilation 2), :133-147 (_make_layer: first block of a layer is NOT dilated)
  heads       a2j/a2j.py:70-89 (depth), :116-135 (regression), :162-181 (classification)
  aggregation a2j/anchor.py:7-42 (anchor grid), :57-82 (post_process.forward)
  wiring      a2j/a2j.py:226-250 (A2JModel.forward / eager_outputs)
All functions take a reference-layout state_dict.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(x, sd, name):
    # nn.BatchNorm2d in eval mode (a2j/resnet.py:67-71; heads a2j/a2j.py:51-60)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], training=False, eps=EPS)


def _bottleneck(x, sd, p, stride, dilation):
    # a2j/resnet.py:78-96; stride and dilation sit on the 3x3 conv (:68)
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=dilation, dilation=dilation)
    out = F.relu(_bn(out, sd, p + "bn2"))
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    if (p + "downsample.0.weight") in sd:
        identity = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1")
    else:
        identity = x
    return F.relu(out + identity)


# (planes, blocks, stride of first block, dilation of blocks 1..)  a2j/resnet.py:109-112
_LAYERS = [(64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 2, 1), (512, 3, 1, 2)]


def backbone(x, sd, channel_in=1):
    """a2j/a2j.py:194-210: x [B,C,H,W] -> (x3 [B,1024,H/16,W/16], x4 [B,2048,H/16,W/16])."""
    n, c, h, w = x.shape
    x = x[:, 0:channel_in]
    if channel_in == 1:
        x = x.expand(n, 3, h, w)
    p = "Backbone.model."
    x = F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(x, sd, p + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = []
    for li, (planes, blocks, stride, dil) in enumerate(_LAYERS, start=1):
        for b in range(blocks):
            x = _bottleneck(x, sd, f"{p}layer{li}.{b}.", stride if b == 0 else 1, 1 if b == 0 else dil)
        feats.append(x)
    return feats[2], feats[3]


def _head(x, sd, name):
    for i in range(1, 5):
        x = F.conv2d(x, sd[f"{name}.conv{i}.weight"], sd[f"{name}.conv{i}.bias"], padding=1)
        x = F.relu(_bn(x, sd, f"{name}.bn{i}"))
    return F.conv2d(x, sd[f"{name}.output.weight"], sd[f"{name}.output.bias"], padding=1)


def heads_raw(x3, x4, sd):
    """Raw NCHW head conv outputs (before the reference's permute/view)."""
    return (_head(x3, sd, "classificationModel"), _head(x4, sd, "regressionModel"),
            _head(x4, sd, "DepthRegressionModel"))


def heads_to_reference_layout(cls_o, reg_o, dep_o, joints=21, anchors=16):
    """The permute(0,3,2,1) + view of a2j/a2j.py:84-89,130-135,176-181."""
    b = cls_o.shape[0]
    c1 = cls_o.permute(0, 3, 2, 1).contiguous().view(b, -1, joints)
    r1 = reg_o.permute(0, 3, 2, 1).contiguous().view(b, -1, joints, 2)
    d1 = dep_o.permute(0, 3, 2, 1).contiguous().view(b, -1, joints)
    return c1, r1, d1


def all_anchors(shape=(11, 11), stride=16, P=(2, 6, 10, 14)) -> torch.Tensor:
    """a2j/anchor.py:7-42 with the A2JModel ctor arguments (a2j/a2j.py:223)."""
    P = np.asarray(P)
    A = len(P) * len(P)
    anchors = np.zeros((A, 2))
    k = 0
    for i in range(len(P)):
        for j in range(len(P)):
            anchors[k, 1] = P[j]
            anchors[k, 0] = P[i]
            k += 1
    shift_h = np.arange(0, shape[0]) * stride
    shift_w = np.arange(0, shape[1]) * stride
    shift_h, shift_w = np.meshgrid(shift_h, shift_w)
    shifts = np.vstack((shift_h.ravel(), shift_w.ravel())).transpose()
    K = shifts.shape[0]
    out = anchors.reshape((1, A, 2)) + shifts.reshape((1, K, 2)).transpose((1, 0, 2))
    return torch.from_numpy(out.reshape((K * A, 2))).float()


def post_process(cls, reg, dep, anchors=None):
    """a2j/anchor.py:57-82: cls [B,N,J], reg [B,N,J,2], dep [B,N,J] -> [B,J,3]."""
    if anchors is None:
        anchors = all_anchors()
    outs = []
    for j in range(cls.shape[0]):
        regj = anchors.unsqueeze(1) + reg[j]
        w = F.softmax(cls[j], dim=0)
        wxy = w.unsqueeze(2).expand(w.shape[0], w.shape[1], 2)
        pxy = (wxy * regj).sum(0)
        pd = (w * dep[j]).sum(0).unsqueeze(1)
        outs.append(torch.cat((pxy, pd), 1))
    return torch.stack(outs)


def a2j_forward(x, sd, channel_in=1, return_heads=False, dtype=torch.float32):
    """a2j/a2j.py:243-250 (gt=None): x [B,1,176,176] metres -> [B,21,3] on CPU.  dtype=torch.float64 (with a state_dict
    converted by `to_dtype`) runs the same operations in double: the yardstick for the fp32 path's own rounding noise."""
    with torch.no_grad():
        x3, x4 = backbone(x.to(dtype), sd, channel_in)
        raw = heads_raw(x3, x4, sd)
        joints = raw[0].shape[1] // 16
        cls, reg, dep = heads_to_reference_layout(*raw, joints=joints)
        anchors = all_anchors((x.shape[2] // 16, x.shape[3] // 16)).to(dtype)
        out = post_process(cls, reg, dep, anchors)
    if return_heads:
        return out, (x3, x4), raw
    return out


def to_dtype(sd, dtype):
    """state_dict with every floating-point tensor converted (integer buffers untouched)"""
    return {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}


def convert_joints(jt_uvd, box, paras, crop_w=176, crop_h=176):
    """a2j/a2j.py:17-34 + datasets3d/a2jdataset.py:31-38: crop-uvd -> camera xyz in mm.  With the detector's int64 box (the live
    caller, ros_demo.py:289) numpy promotes the reference's arithmetic to float64; with the DATASET's float32 box and intrinsics
    (the evaluation caller, a2j/a2j.py:339-346; a2jdataset.py:293) every operation stays in float32, in the reference's order."""
    if np.asarray(box).dtype == np.float32:
        jt = np.asarray(jt_uvd, dtype=np.float32).reshape(-1, 3)
        b = np.asarray(box).reshape(4)
        out = np.ones_like(jt)
        out[:, 0] = jt[:, 0] * (b[2] - b[0]) / np.float32(crop_w) + b[0]
        out[:, 1] = jt[:, 1] * (b[3] - b[1]) / np.float32(crop_h) + b[1]
        out[:, 2] = jt[:, 2]
        if paras is not None:
            p = np.asarray(paras, dtype=np.float32).reshape(4)
            out[:, :2] = (out[:, :2] - p[2:]) * out[:, 2:] / p[:2]
            out = out * np.float32(1000.0)
        return out
    jt = np.asarray(jt_uvd, dtype=np.float64).reshape(-1, 3)
    x0, y0, x1, y1 = [float(v) for v in np.asarray(box).reshape(4)]
    out = np.ones_like(jt)
    out[:, 0] = jt[:, 0] * (x1 - x0) / crop_w + x0
    out[:, 1] = jt[:, 1] * (y1 - y0) / crop_h + y0
    out[:, 2] = jt[:, 2]
    if paras is not None:
        fx, fy, cx, cy = [float(v) for v in paras]
        out[:, 0] = (out[:, 0] - cx) * out[:, 2] / fx
        out[:, 1] = (out[:, 1] - cy) * out[:, 2] / fy
        out = out * 1000.0
    return out.astype(np.float32)

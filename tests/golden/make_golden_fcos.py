"""Golden vectors for the FCOS in-repo code and the HandNet glue, produced by running the
reference's own classes (fcos_utils/fcos.py, det_utils.py, anchor_utils.py,
handnet_pipeline/handnet_pipeline.py) in the build container.

torchvision is not installed, so the names the reference imports from it are bound to
stand-ins backed by the oracle's restatement (oracle/fcos_ref.py): the transform, the
ResNet-34-FPN backbone and batched_nms.  What these goldens therefore PIN is the
reference's in-repo arithmetic and wiring (heads incl. GroupNorm, anchors, box decode,
score/threshold/compaction, dict assembly, box rescale, HandNet crop logic); the
torchvision pieces stay "parity unpinned" (see oracle/__init__.py).
Imported by make_golden.py; never runs on the GPU box.
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch
from torch import nn

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
sys.path.insert(0, str(REPO))

from hn_amd import synth  # noqa: E402
from oracle import fcos_ref  # noqa: E402

_FCOS_SD = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes


class _Transform(nn.Module):
    def __init__(self, min_size, max_size, image_mean, image_std):
        super().__init__()
        assert (min_size, max_size) == (800, 1333)
        assert list(image_mean) == fcos_ref.IMAGE_MEAN and list(image_std) == fcos_ref.IMAGE_STD

    def forward(self, images, targets=None):
        t, sizes = fcos_ref.transform(images)
        return _ImageList(t, sizes), targets


class _Backbone(nn.Module):
    out_channels = 256

    def forward(self, x):
        return fcos_ref.backbone(x, _FCOS_SD["sd"])


def install_fcos_shim():
    from make_golden import install_a2j_shim
    install_a2j_shim()
    # the reference must win over this repo's same-named drop-in packages
    while str(REF) in sys.path:
        sys.path.remove(str(REF))
    sys.path.insert(0, str(REF))
    tv = sys.modules["torchvision"]
    boxes_ns = types.SimpleNamespace(batched_nms=fcos_ref.batched_nms)
    tv.ops = _mod("torchvision.ops", sigmoid_focal_loss=None, boxes=boxes_ns)
    _mod("torchvision.ops.misc", FrozenBatchNorm2d=nn.Module)
    _mod("torchvision.ops.feature_pyramid_network", LastLevelP6P7=None)
    _mod("torchvision.models")
    _mod("torchvision.models.detection")
    _mod("torchvision.models.detection.transform", GeneralizedRCNNTransform=_Transform)
    _mod("torchvision.models.detection.backbone_utils",
         resnet_fpn_backbone=lambda name, pretrained=True, returned_layers=None: _Backbone())
    _mod("torchvision.models.detection.image_list", ImageList=_ImageList)
    # the reference does `from matplotlib.pyplot import box` (fcos.py:6); avoid importing a GUI stack
    if "matplotlib" not in sys.modules:
        _mod("matplotlib")
        _mod("matplotlib.pyplot", box=None)


def gen_fcos():
    install_fcos_shim()
    from fcos_utils.fcos import FCOS
    sd = synth.make_fcos_state_dict(seed=0, num_classes=3)
    _FCOS_SD["sd"] = sd
    det = FCOS(num_classes=3, ext=False, nms_thresh=0.5).eval()
    missing, unexpected = det.load_state_dict(sd, strict=False)
    assert not missing, missing
    assert all(k.startswith("backbone.") for k in unexpected)
    rgb = synth.make_rgb(1, seed=1000)
    images = [rgb[0]]
    with torch.inference_mode():
        # intermediate head tensors through the reference's own head
        il, _ = det.transform(images, None)
        feats = list(det.backbone(il.tensors).values())[:-1]
        ho = det.head(feats)
        anchors = det.anchor_generator(il, feats)
        out = det(images, None)
    d = out[0]
    k = d["boxes"].shape[0]
    print("fcos golden: detections", k, "labels", np.bincount(d["labels"].numpy(), minlength=3))
    # candidate count before NMS, for information
    sc = torch.sqrt(torch.sigmoid(ho["cls_logits"]) * torch.sigmoid(ho["bbox_ctrness"])).max(-1)[0]
    print("candidates > 0.7:", int((sc > 0.7).sum()))
    np.savez_compressed(
        HERE / "fcos_forward.npz", rgb_seed=np.int64(1000), weight_seed=np.int64(0),
        boxes=d["boxes"].numpy(), scores=d["scores"].numpy(), labels=d["labels"].numpy(),
        sides=d["sides"].numpy(), feature_idx=d["feature_idx"].numpy(),
        n_candidates=np.int64(int((sc > 0.7).sum())),
        cls_probe=ho["cls_logits"][0, ::97].numpy(), reg_probe=ho["bbox_regression"][0, ::97].numpy(),
        ctr_probe=ho["bbox_ctrness"][0, ::97].numpy(), lr_probe=ho["hand_lr"][0, ::97].numpy(),
        anchors_probe=anchors[0][::97].numpy(), num_anchors=np.int64(anchors[0].shape[0]),
    )


def gen_fcos_ext():
    """ext=True detector (the class default; trainval_net_fcos.py --test-only): contact-state / dxdy outputs."""
    install_fcos_shim()
    from fcos_utils.fcos import FCOS
    sd = synth.make_fcos_state_dict(seed=0, num_classes=3, ext=True)
    _FCOS_SD["sd"] = sd
    det = FCOS(num_classes=3, ext=True, nms_thresh=0.5).eval()
    missing, unexpected = det.load_state_dict(sd, strict=False)
    assert not missing, missing
    rgb = synth.make_rgb(1, seed=1000)
    with torch.inference_mode():
        d = det([rgb[0]], None)[0]
    assert set(d) == {"boxes", "scores", "labels", "dxdymags", "contacts", "sides"}, set(d)
    print("fcos ext golden: detections", d["boxes"].shape[0], "contacts", np.bincount(d["contacts"].numpy(), minlength=5))
    np.savez_compressed(
        HERE / "fcos_ext_forward.npz", rgb_seed=np.int64(1000), weight_seed=np.int64(0),
        boxes=d["boxes"].numpy(), scores=d["scores"].numpy(), labels=d["labels"].numpy(),
        sides=d["sides"].numpy(), contacts=d["contacts"].numpy(), dxdymags=d["dxdymags"].numpy(),
    )


def gen_handnet():
    install_fcos_shim()
    from handnet_pipeline.handnet_pipeline import HandNet
    fsd = synth.make_fcos_state_dict(seed=0, num_classes=3)
    asd = synth.make_a2j_state_dict(seed=0)
    _FCOS_SD["sd"] = fsd
    args = types.SimpleNamespace(pretrained_fcos="none.pth", pretrained_a2j="none.pth")
    net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=False, RGBD=False).eval()
    net.detector.load_state_dict(fsd, strict=False)
    net.a2j.load_state_dict(asd, strict=False)
    rgb = synth.make_rgb(2, seed=1000)
    depth = synth.make_depth(2, seed=2000)
    with torch.inference_mode():
        kp, depth_batch, crops = net([rgb[0], rgb[1]], depth_images=depth)
        none_out = net([rgb[0]], depth_images=depth[:1], is_detect=True)
    assert none_out is None
    print("handnet golden: crops", crops.tolist(), "kp[0,0]", kp[0, 0].tolist())
    np.savez_compressed(
        HERE / "handnet_forward.npz", rgb_seed=np.int64(1000), depth_seed=np.int64(2000),
        keypoints=kp.numpy(), crops=crops.numpy(),
        depth_batch_probe=depth_batch[:, 0, ::16, ::16].numpy(),
        depth_batch_sum=depth_batch.double().sum().item(),
    )


def gen_handnet_rgbd():
    """RGBD=True glue (handnet_pipeline.py:101-102: 4-channel crop + channel permutation [2,1,0,3]).
    The reference builds this model through a Lightning checkpoint (handnet_pipeline.py:28-29); offline the
    same A2JModel(is_RGBD=True) is constructed directly and given the synthetic RGBD weights."""
    install_fcos_shim()
    from handnet_pipeline.handnet_pipeline import HandNet
    from a2j.a2j import A2JModel
    fsd = synth.make_fcos_state_dict(seed=0, num_classes=3)
    asd = synth.make_a2j_state_dict(seed=0, rgbd=True)
    _FCOS_SD["sd"] = fsd
    args = types.SimpleNamespace(pretrained_fcos="none.pth", pretrained_a2j="none.pth")
    net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=False, RGBD=False).eval()
    net.detector.load_state_dict(fsd, strict=False)
    net.RGBD = True
    net.a2j = A2JModel(21, crop_height=176, crop_width=176, is_RGBD=True).eval()
    net.a2j.load_state_dict(asd, strict=False)
    rgb = synth.make_rgb(2, seed=1000)
    depth = synth.make_depth(2, seed=2000)
    rgbd = torch.cat([rgb, depth], dim=1)                     # ros_demo.py:268-270 (RGB + depth)
    with torch.inference_mode():
        kp, depth_batch, crops = net([rgb[0], rgb[1]], depth_images=rgbd)
    print("handnet rgbd golden: crops", crops.tolist(), "kp[0,0]", kp[0, 0].tolist())
    np.savez_compressed(
        HERE / "handnet_rgbd_forward.npz", rgb_seed=np.int64(1000), depth_seed=np.int64(2000),
        keypoints=kp.numpy(), crops=crops.numpy(),
        depth_batch_probe=depth_batch[:, :, ::16, ::16].numpy(),
        depth_batch_sum=depth_batch.double().sum(dim=(0, 2, 3)).numpy(),
    )

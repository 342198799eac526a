#!/usr/bin/env python3 -B
"""Golden vectors for convert_joints + uvd2xyz by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  The functions under test are the reference's own
`a2j.a2j.convert_joints` (a2j/a2j.py:17-43) and `datasets3d.a2jdataset.uvd2xyz` (datasets3d/a2jdataset.py:31-38);
both are pure numpy.  Only imports that do no arithmetic here are stubbed (dataset toolkit, MANO layer, PIL, cv2,
pycocotools, Lightning, visualisation, torchvision.transforms).

    python -B tests/golden/make_golden_joints.py      ->  tests/golden/convert_joints.npz
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
sys.dont_write_bytecode = True


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    _mod("dex_ycb_toolkit")
    _mod("dex_ycb_toolkit.factory", get_dataset=None)
    _mod("dex_ycb_toolkit.hpe_eval", HPEEvaluator=object)
    _mod("manopth")
    _mod("manopth.manolayer", ManoLayer=None)
    _mod("PIL", Image=None)
    _mod("pycocotools")
    sys.modules["pycocotools"].mask = _mod("pycocotools.mask")
    _mod("cv2")
    _mod("pytorch_lightning", LightningModule=torch.nn.Module, LightningDataModule=object)
    _mod("utils")
    _mod("utils.utils", get_e2e_loaders=None, vis_minibatch=None)
    _mod("utils.vistool", VisualUtil=None)
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    sys.path.insert(0, str(REF))
    import a2j.resnet as R
    orig = R.resnet50
    R.resnet50 = lambda pretrained=False, **kw: orig(pretrained=False, **kw)
    from datasets3d.a2jdataset import uvd2xyz, xyz2uvd       # the reference's own (real module, stubbed imports)
    from a2j.a2j import convert_joints                        # binds the real uvd2xyz above

    rng = np.random.RandomState(20261004)
    cases = 12
    pred = np.zeros((cases, 21, 3), np.float32)
    gt = np.zeros((cases, 21, 3), np.float32)
    box = np.zeros((cases, 4), np.int64)
    paras = np.zeros((cases, 4), np.float32)
    xyz_pred = np.zeros((cases, 21, 3), np.float32)
    xyz_gt = np.zeros((cases, 21, 3), np.float32)
    uvd_img = np.zeros((cases, 21, 3), np.float32)
    for i in range(cases):
        pred[i, :, :2] = rng.uniform(-8.0, 184.0, (21, 2)).astype(np.float32)     # crop pixels, slightly outside too
        pred[i, :, 2] = rng.uniform(0.25, 1.6, 21).astype(np.float32)             # metres
        gt[i] = pred[i] + rng.normal(0, 0.5, (21, 3)).astype(np.float32) * np.array([1, 1, 0.01], np.float32)
        x1, y1 = rng.randint(0, 400), rng.randint(0, 300)
        box[i] = [x1, y1, x1 + rng.randint(1, 640 - x1 + 1), y1 + rng.randint(1, 480 - y1 + 1)]
        paras[i] = [617.343, 617.343, 312.42, 241.42] if i % 2 == 0 else \
            [rng.uniform(400, 900), rng.uniform(400, 900), rng.uniform(280, 360), rng.uniform(200, 280)]
        a, b = convert_joints(pred[i].copy(), gt[i].copy(), box[i].copy(), paras[i].copy(), 176, 176)
        xyz_pred[i], xyz_gt[i] = a, b
        uvd_img[i] = convert_joints(pred[i].copy(), None, box[i].copy(), None, 176, 176)   # paras=None branch
    # uvd2xyz / xyz2uvd on their own (flipy default)
    pts = rng.uniform(-0.4, 0.4, (16, 21, 3)).astype(np.float32)
    pts[..., 2] = rng.uniform(0.3, 1.5, (16, 21)).astype(np.float32)
    p0 = np.array([617.343, 617.343, 312.42, 241.42], np.float32)
    uvd = xyz2uvd(pts, p0)
    back = uvd2xyz(uvd, p0)
    # the evaluation caller's form (A2JModelLightning.test_step, a2j/a2j.py:333-348): the DATASET's box, float32 with fractional
    # corners (a2jdataset.py:262-265,293), and the sample's own intrinsics (:279,293) -- every operation stays in float32.
    # (Drawn after everything above: the earlier arrays of the file do not change.)
    box_f32 = np.zeros((cases, 4), np.float32)
    xyz_pred_f32 = np.zeros((cases, 21, 3), np.float32)
    xyz_gt_f32 = np.zeros((cases, 21, 3), np.float32)
    uvd_img_f32 = np.zeros((cases, 21, 3), np.float32)
    for i in range(cases):
        x1, y1 = rng.uniform(0, 400), rng.uniform(0, 300)
        box_f32[i] = [x1, y1, x1 + rng.uniform(20, 639 - x1), y1 + rng.uniform(20, 479 - y1)]
        a, b = convert_joints(pred[i].copy(), gt[i].copy(), box_f32[i].copy(), paras[i].copy(), 176, 176)
        assert a.dtype == np.float32 and b.dtype == np.float32
        xyz_pred_f32[i], xyz_gt_f32[i] = a, b
        uvd_img_f32[i] = convert_joints(pred[i].copy(), None, box_f32[i].copy(), None, 176, 176)
    np.savez_compressed(HERE / "convert_joints.npz", pred=pred, gt=gt, box=box, paras=paras, xyz_pred=xyz_pred,
                        xyz_gt=xyz_gt, uvd_img=uvd_img, pts=pts, p0=p0, uvd=uvd, back=back, box_f32=box_f32,
                        xyz_pred_f32=xyz_pred_f32, xyz_gt_f32=xyz_gt_f32, uvd_img_f32=uvd_img_f32)
    print("wrote", HERE / "convert_joints.npz", "max |xyz|", float(np.abs(xyz_pred).max()))


if __name__ == "__main__":
    main()

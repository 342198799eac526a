#!/usr/bin/env python3 -B
"""Golden vectors for the live caller's glue between the pose network and the lifter (ros_demo.py:148-157), produced by
IMPORTING the reference's pose2mesh/lib/coord_utils.py and aug_utils.py in the build container.

What the imported reference computes here: get_bbox, process_bbox, get_center_scale (coord_utils.py:7-66) and
affine_transform (aug_utils.py:176-179) -- pure numpy.  What it cannot compute here: get_affine_transform's last line calls
cv2.getAffineTransform, and OpenCV is not installed (no wheel, no source on the image); that one call is the exact solution of
a three-point correspondence and is restated as such in oracle/pose2mesh_ref.py (definition-anchored, not pinned).

Stubs are non-arithmetic only: core.config (cfg.MODEL.input_shape = (384, 288), its default at core/config.py:52; the
reference's module creates experiment directories at import time) and an EMPTY cv2 module (aug_utils imports it at the top).
    python -B tests/golden/make_golden_lifter_input.py
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/pose2mesh/lib")
sys.dont_write_bytecode = True


def install_shim():
    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    cfg = EasyDict(MODEL=EasyDict(input_shape=(384, 288)))          # core/config.py:52
    core = types.ModuleType("core")
    core.__path__ = []
    conf = types.ModuleType("core.config")
    conf.cfg = cfg
    core.config = conf
    sys.modules["core"], sys.modules["core.config"] = core, conf
    sys.modules["cv2"] = types.ModuleType("cv2")                     # (empty: nothing of it is called below)
    sys.path.insert(0, str(REF))


def main():
    install_shim()
    import aug_utils
    import coord_utils
    rng = np.random.default_rng(20)
    joints, bbox, bbox2, ok, center, scale = [], [], [], [], [], []
    for i in range(24):
        if i == 20:
            j = np.full((21, 2), 7.0, dtype=np.float32)                               # degenerate: process_bbox returns None
        elif i == 21:
            j = np.tile(np.array([[50.0, 60.0]], dtype=np.float32), (21, 1))
            j[:, 0] += np.arange(21, dtype=np.float32)                                # zero height
        else:
            j = (rng.uniform(40, 600, 2) + rng.normal(size=(21, 2)) * rng.uniform(3, 140, 2)).astype(np.float32)
        b = coord_utils.get_bbox(j)
        b2 = coord_utils.process_bbox(b.copy())
        joints.append(j)
        bbox.append(b)
        ok.append(b2 is not None)
        bbox2.append(np.zeros(4) if b2 is None else np.asarray(b2, dtype=np.float64))
        if b2 is None:
            center.append(np.zeros(2, np.float32)); scale.append(np.zeros(2, np.float32))
        else:
            c, s = coord_utils.get_center_scale(b2)
            center.append(c); scale.append(s)
    # affine_transform(pt, t) for a few 2 x 3 maps
    ts = rng.normal(size=(6, 2, 3))
    pts = rng.uniform(0, 640, size=(6, 5, 2))
    warped = np.stack([np.stack([aug_utils.affine_transform(p, t) for p in ps]) for ps, t in zip(pts, ts)])
    np.savez_compressed(HERE / "lifter_input.npz", joints=np.stack(joints), bbox=np.stack(bbox), bbox2=np.stack(bbox2),
                        ok=np.asarray(ok), center=np.stack(center), scale=np.stack(scale), t=ts, pts=pts, warped=warped,
                        input_shape=np.asarray([384, 288]))
    print("wrote", HERE / "lifter_input.npz", "cases", len(joints), "rejected", int((~np.asarray(ok)).sum()))


if __name__ == "__main__":
    main()

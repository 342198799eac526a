"""Drop-in `a2j.a2j.A2JModel` running on MI355X HIP kernels.

Same constructor, state_dict layout, call signature and return convention as the
reference class (a2j/a2j.py:212-250): `model(x[K,1,176,176]) -> FloatTensor[K,21,3]` on
the CPU (the reference ends with `.data.cpu()`, a2j/a2j.py:229).  Callers keep writing

    model = A2JModel(21, crop_height=176, crop_width=176, is_RGBD=False).cuda().eval()
    model.load_state_dict(torch.load(path, map_location="cpu")["model"], strict=False)
    jt_uvd = model(depth)[0]                                  # a2j_infer.py:25-28,58-60

Training (`gt is not None`) is out of scope and raises.
"""
from __future__ import annotations

import numpy as np
import torch

from hn_amd import synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd.state import EngineOwner, build_state_tree


class A2JModel(EngineOwner):
    def __init__(self, num_classes, crop_height, crop_width, is_3D=True, is_RGBD=False, spatial_factor=0.5):
        super().__init__()
        if not is_3D:
            raise NotImplementedError("only the is_3D=True model of the reference's inference path is provided")
        self.is_3D = is_3D
        self.is_RGBD = is_RGBD
        self.num_classes = num_classes
        self.crop_height, self.crop_width = crop_height, crop_width
        # The reference starts from ImageNet weights fetched over the network
        # (a2j/a2j.py:188); offline, the state starts from the deterministic synthetic
        # checkpoint and is normally overwritten by load_state_dict().
        tree = build_state_tree(synth.make_a2j_state_dict(seed=0, num_joints=num_classes, rgbd=is_RGBD))
        for name, child in tree.named_children():
            self.add_module(name, child)

    def engine(self) -> A2JEngine:
        dev = self._require_gpu()
        if self._engine is None:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._engine = A2JEngine(sd, num_joints=self.num_classes, rgbd=self.is_RGBD, device=dev)
        return self._engine

    def forward_device(self, x, valid=None):
        """Same as forward() but leaves the [K,J,3] result on the GPU (no forced sync)."""
        return self.engine().forward(x, valid)

    def forward(self, x, gt=None):
        if gt is not None:
            raise NotImplementedError("training losses (a2j/anchor.py:84-152) are outside the inference hot path")
        return self.forward_device(x).cpu()


def convert_joints(jt_uvd_pred, jt_uvd_gt, box, paras, cropWidth, cropHeight):
    """crop-(u,v,d) -> image (u,v,d) -> camera xyz in mm; same contract as a2j/a2j.py:17-43."""
    def one(jt):
        jt = np.asarray(jt).reshape(-1, 3)
        b = np.asarray(box).reshape(4)
        out = np.ones_like(jt)
        out[:, 0] = jt[:, 0] * (b[2] - b[0]) / cropWidth + b[0]
        out[:, 1] = jt[:, 1] * (b[3] - b[1]) / cropHeight + b[1]
        out[:, 2] = jt[:, 2]
        if paras is not None:
            p = np.asarray(paras).reshape(4)
            out[:, :2] = (out[:, :2] - p[2:]) * out[:, 2:] / p[:2]
            out = out.astype(np.float32) * 1000.0
        return out

    pred = one(jt_uvd_pred)
    if jt_uvd_gt is not None:
        return pred, one(jt_uvd_gt)
    return pred

"""Drop-in `models.pose2mesh_net` (pose2mesh/lib/models/pose2mesh_net.py:9-30) on MI355X.

    model = models.pose2mesh_net.get_model(joint_num, graph_L)          # ros_demo.py:142
    model.load_state_dict(checkpoint['model_state_dict'])               # ros_demo.py:144
    pred_mesh, pose3d = model(joint_img)                                # ros_demo.py:160, joint_img [B,21,2] on the GPU

Same constructor arguments and state_dict layout (pose_lifter.*, pose2mesh.{fc,cl.N,bn.N}.*) as the reference's
FlatPose2Mesh; graph_L is the list of rescaled Laplacians graph_utils.build_coarse_graphs returns (scipy sparse,
finest first, joint graph last).  Both outputs stay on the device, like the reference's.  Inference ('mano'
configuration, eval) only; there is no CPU fallback.
"""
from __future__ import annotations

import os
import sys

import torch

_PKG = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))   # handnet-pipeline_amd/
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from hn_amd import synth  # noqa: E402
from hn_amd.pose2mesh_engine import Pose2MeshEngine  # noqa: E402
from hn_amd.state import EngineOwner, build_state_tree  # noqa: E402


class FlatPose2Mesh(EngineOwner):
    def __init__(self, num_joint, graph_L):
        super().__init__()
        self.num_joint = num_joint
        self._graph_L = list(graph_L)
        sizes = [int(L.shape[0]) for L in self._graph_L]
        tree = build_state_tree(synth.make_pose2mesh_state_dict(seed=0, graph_sizes=sizes, num_joint=num_joint))
        for name, child in tree.named_children():
            self.add_module(name, child)

    def engine(self) -> Pose2MeshEngine:
        dev = self._require_gpu()
        if self._engine is None:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._engine = Pose2MeshEngine(sd, self._graph_L, num_joint=self.num_joint, device=dev)
        return self._engine

    def forward(self, pose2d):
        if self.training:
            raise NotImplementedError("training (dropout / batch statistics) is outside the inference path")
        return self.engine().forward(pose2d)


def get_model(num_joint, graph_L):
    return FlatPose2Mesh(num_joint, graph_L)

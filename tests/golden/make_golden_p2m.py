#!/usr/bin/env python3 -B
"""Golden vectors for the Pose2Mesh lifter, produced by IMPORTING the reference's
pose2mesh/lib modules (models.pose2mesh_net, graph_utils, coarsening) in the build container.

The real graph hierarchy is derived from the MANO model files, which are neither in the reference tree nor in
this image.  The reference's own graph code (graph_utils.build_coarse_graphs -> coarsening.coarsen) is therefore
run on a SYNTHETIC 778-vertex triangle mesh (seeded Delaunay triangulation; MANO has 778 vertices / 1538 faces),
with the joint graph exactly as ros_demo.py:128-136 builds it.  What this pins is the arithmetic of
FlatPose2Mesh / LinearModel / Pose2Mesh / graph_conv_cheby for a hierarchy of the real sizes; the Laplacians
travel as data (CSR arrays) inside the fixture because the reference cannot run on the GPU box.

Stubs are non-arithmetic only: core.config (the reference's module creates experiment directories next to
itself at import time, which must not happen in the read-only reference tree; only cfg.DATASET.target_joint_set
= 'mano' and cfg.MODEL.posenet_pretrained = False, its defaults at core/config.py:45,56, are consumed by the
models) and funcs_utils (checkpoint loader, imports cv2).
    python -B tests/golden/make_golden_p2m.py
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference/pose2mesh/lib")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
sys.path.insert(0, str(REPO))

from hn_amd import synth  # noqa: E402

SKELETON = ((0, 1), (0, 5), (0, 9), (0, 13), (0, 17), (1, 2), (2, 3), (3, 4), (5, 6), (6, 7), (7, 8), (9, 10), (10, 11),
            (11, 12), (13, 14), (14, 15), (15, 16), (17, 18), (18, 19), (19, 20))            # ros_demo.py:128
HORI_CONN = ((1, 5), (5, 9), (9, 13), (13, 17), (2, 6), (6, 10), (10, 14), (14, 18), (3, 7), (7, 11), (11, 15), (15, 19),
             (4, 8), (8, 12), (12, 16), (16, 20))                                            # ros_demo.py:129-131


def install_shim():
    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    cfg = EasyDict(DATASET=EasyDict(target_joint_set="mano"),                     # core/config.py:45
                   MODEL=EasyDict(posenet_pretrained=False, posenet_path=""))      # core/config.py:56
    core = types.ModuleType("core")
    core.__path__ = []
    conf = types.ModuleType("core.config")
    conf.cfg = cfg
    core.config = conf
    sys.modules["core"], sys.modules["core.config"] = core, conf
    f = types.ModuleType("funcs_utils")
    f.load_checkpoint = None
    sys.modules["funcs_utils"] = f
    # the repo's own drop-in packages must not shadow the reference's `models`
    sys.path.insert(0, str(REF))


def synthetic_faces(seed=7, nv=778):
    from scipy.spatial import Delaunay
    pts = np.random.default_rng(seed).random((nv, 2))
    return Delaunay(pts).simplices.astype(np.int64)


def main():
    install_shim()
    import graph_utils
    from models import pose2mesh_net
    faces = synthetic_faces()
    assert faces.max() + 1 == 778
    _, graph_L, graph_perm, perm_rev = graph_utils.build_coarse_graphs(faces, 21, SKELETON, HORI_CONN, levels=6)
    sizes = [L.shape[0] for L in graph_L]
    print("graph sizes:", sizes, "nnz:", [L.nnz for L in graph_L])
    fixture = {}
    for i, L in enumerate(graph_L):
        c = sp.csr_matrix(L).astype(np.float32)
        c.sort_indices()
        fixture[f"L{i}_indptr"], fixture[f"L{i}_indices"], fixture[f"L{i}_data"] = c.indptr, c.indices, c.data
        fixture[f"L{i}_shape"] = np.array(c.shape)
    keep = [sp.csr_matrix(L) for L in graph_L]
    model = pose2mesh_net.get_model(21, list(graph_L)).eval()     # the ctor deletes graph_L[-2] and converts in place
    sd = synth.make_pose2mesh_state_dict(seed=0, graph_sizes=[k.shape[0] for k in keep])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("num_batches_tracked" in k for k in missing), missing
    g = torch.Generator().manual_seed(5000)
    pose2d = torch.randn((3, 21, 2), generator=g)
    model.pose2mesh.graph_L = [L.cpu() for L in model.pose2mesh.graph_L]
    torch.Tensor.cuda = lambda self, *a, **k: self          # Pose2Mesh.forward moves graph_L with .cuda() (meshnet.py:87)
    with torch.inference_mode():
        mesh, pose3d = model(pose2d)
    print("mesh", tuple(mesh.shape), float(mesh.abs().mean()), "pose3d", tuple(pose3d.shape), float(pose3d.abs().mean()))
    np.savez_compressed(HERE / "pose2mesh_forward.npz", num_levels=np.int64(len(keep)), input_seed=np.int64(5000),
                        weight_seed=np.int64(0), mesh=mesh.numpy(), pose3d=pose3d.numpy(),
                        perm_reverse=np.asarray(perm_rev, dtype=np.int64), **fixture)


if __name__ == "__main__":
    main()

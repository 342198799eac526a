"""Pose2Mesh lifter on HIP vs the reference golden and the oracle."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def p2m(golden_dir):
    from hn_amd import synth
    from oracle import pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    sd = synth.make_pose2mesh_state_dict(seed=int(g["weight_seed"]), graph_sizes=[m.shape[0] for m in graphs])
    return g, graphs, sd


def test_graph_ops_match_torch(p2m):
    """x1 = L x0, the k-major S32 basis [x0 | x1 | 2 L x1 - x0 | 0] and the feature-axis residual, vs torch CPU ops."""
    import torch.nn.functional as F
    from hn_amd import ops
    from oracle import pose2mesh_ref
    _, graphs, _ = p2m
    L = graphs[3]
    g = ops.csr_graph(L, "cuda")
    Lt = pose2mesh_ref._to_torch_sparse(L)
    x = torch.randn((2, L.shape[0], 24), generator=torch.Generator().manual_seed(1))
    x1 = torch.stack([torch.sparse.mm(Lt, x[b]) for b in range(2)])
    x2 = torch.stack([2 * torch.sparse.mm(Lt, x1[b]) - x[b] for b in range(2)])
    y1 = ops.spmm_csr(g, x.cuda())
    assert (y1.cpu() - x1).abs().max().item() <= 1e-6 * max(1.0, x1.abs().max().item())
    basis = ops.from_split(ops.cheby3_basis_split(g, x.cuda(), y1)).cpu()      # [2, V, 1, 96]
    want = torch.cat([x, x1, x2], dim=2)
    assert basis.shape == (2, L.shape[0], 1, 96)
    assert (basis[:, :, 0, :72] - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    assert float(basis[:, :, 0, 72:].abs().max()) == 0.0
    xin = torch.randn((2, 50, 64), generator=torch.Generator().manual_seed(2))
    y = torch.randn((2, 50, 256), generator=torch.Generator().manual_seed(3))
    ref = F.interpolate(xin, size=256, mode="linear") + y
    ref = F.interpolate(ref.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    got = ops.feat_interp_add(xin.cuda(), y.cuda(), up=2).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 1e-5
    same = ops.feat_interp_add(y.cuda(), y.cuda(), up=1).cpu()                  # equal sizes: identity + add
    assert torch.equal(same, y + y)


def test_pose2mesh_dropin_matches_reference_golden(p2m):
    """models.pose2mesh_net.get_model(...) as ros_demo.py:142-160 uses it, vs the imported reference's outputs.
    Tolerances: pose3d is in millimetres (|x| ~ 1e2): 2e-2; mesh coordinates (|x| ~ 4): 1e-3."""
    g, graphs, sd = p2m
    sys.path.insert(0, str(REPO / "handnet-pipeline_amd" / "pose2mesh" / "lib"))
    try:
        import models
    finally:
        sys.path.pop(0)
    model = models.pose2mesh_net.get_model(21, graphs)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    model = model.cuda().eval()
    pose2d = torch.randn((3, 21, 2), generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.inference_mode():
        mesh, pose3d = model(pose2d.cuda())
    assert mesh.is_cuda and pose3d.is_cuda and mesh.shape == (3, graphs[0].shape[0], 3) and pose3d.shape == (3, 21, 3)
    assert np.abs(pose3d.cpu().numpy() - g["pose3d"]).max() <= 2e-2
    assert np.abs(mesh.cpu().numpy() - g["mesh"]).max() <= 1e-3
    # ros_demo.py:161: vertices of the real mesh in original order
    rev = torch.from_numpy(g["perm_reverse"][:778]).cuda()
    assert mesh[:, rev, :].shape == (3, 778, 3)
    with pytest.raises(RuntimeError):
        model.engine().forward(pose2d)          # CPU tensor: no fallback
    # hipGraph replay of the launch-bound forward reproduces the eager result, also for new inputs
    run, s_in, (g_mesh, g_pose) = model.engine().graphed(pose2d.cuda())
    other = torch.randn((3, 21, 2), generator=torch.Generator().manual_seed(9)).cuda()
    s_in.copy_(other)
    run()
    e_mesh, e_pose = model.engine().forward(other)
    torch.cuda.synchronize()
    # (capture splits more k loops than eager mode does: equal to fp32 rounding, not bitwise)
    assert (g_mesh - e_mesh).abs().max().item() < 1e-4 and (g_pose - e_pose).abs().max().item() < 2e-3

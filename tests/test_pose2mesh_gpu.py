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


def test_fused_graph_conv_matches_the_layer_by_layer_form(p2m):
    """hn_graph_conv_cheby3_f16x3 (basis gather, MFMA, bias / ReLU, block residual + vertex up-sampling in ONE launch) against
    spmm -> basis -> 1x1 convolution -> feat_interp_add on every shape class of the mesh net: narrow input (8 features),
    wide (256), the 3-channel output layer without ReLU, the S32 output in front of fc, batch 1 and 3 (a tile spanning
    two samples), on the coarsest and the finest graph."""
    from hn_amd import ops
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    _, graphs, sd = p2m
    eng = Pose2MeshEngine(sd, graphs, device="cuda")
    gen = torch.Generator().manual_seed(21)
    # (layer index in eng.cl, graph level, residual?, up, out_split)
    cases = [(0, -1, False, 1, False), (2, -1, False, 1, True), (4, -2, True, 2, False), (6, -3, True, 2, False),
             (12, 0, True, 1, False), (13, 0, False, 1, False), (14, 0, False, 1, False)]
    for b in (1, 3):
        for li, lv, res, up, split in cases:
            cw, fin_pad, relu = eng.cl[li]
            g, g2 = eng.graphs[lv], eng.graphs2[lv]
            x = torch.randn((b, g.v, fin_pad), generator=gen).cuda()
            fi = 64 if fin_pad >= 64 else fin_pad
            xin = torch.randn((b, g.v, fi), generator=gen).cuda() if res else None
            got = ops.graph_conv_cheby3(g, g2, x, cw, relu=relu, xin=xin, up=up, out_split=split)
            basis = ops.cheby3_basis_split(g, x, ops.spmm_csr(g, x))
            want = ops.conv2d_nhwc(basis, cw.w, cw.bias, relu=relu, w16=cw.w16, splitk=False).view(b, g.v, -1)
            if res:
                want = ops.feat_interp_add(xin, want.contiguous(), up=up)
            if split:
                got = ops.from_split(got).view(b, g.v * up, -1)
            assert got.shape == want.shape, (li, got.shape, want.shape)
            # the filter bank in the standard layout (no w_frag) takes the same values in the same order: bit-identical
            import types
            plain = types.SimpleNamespace(w=cw.w, bias=cw.bias, w16=cw.w16)
            again = ops.graph_conv_cheby3(g, g2, x, plain, relu=relu, xin=xin, up=up, out_split=split)
            assert torch.equal(ops.from_split(again).view(b, g.v * up, -1) if split else again, got)
            scale = max(1.0, want.abs().max().item())
            assert (got - want).abs().max().item() <= 3e-6 * scale, (b, li, (got - want).abs().max().item(), scale)


def test_fused_lifter_matches_the_layer_by_layer_lifter(p2m):
    """The 26-launch forward (fused=True) against the round-2 launch structure (fused=False) of the same weights: pose3d
    (millimetres, |x| ~ 1e2) within 5e-3, mesh (|x| ~ 4) within 2e-4 -- summation order only (batch_norm2 folded into the
    Linear in front of it, 2 L L - I formed on the host, no split-K in the fused graph convolutions)."""
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    g, graphs, sd = p2m
    a = Pose2MeshEngine(sd, graphs, device="cuda", fused=True)
    b = Pose2MeshEngine(sd, graphs, device="cuda", fused=False)
    for n in (1, 5):
        pose2d = torch.randn((n, 21, 2), generator=torch.Generator().manual_seed(31 + n)).cuda()
        m1, p1 = a.forward(pose2d)
        m2, p2 = b.forward(pose2d)
        assert (p1 - p2).abs().max().item() < 5e-3 and (m1 - m2).abs().max().item() < 2e-4


def test_linear_rows_matches_fp64(p2m):
    """hn_linear_rows_f16x3 (a Linear on 1..4 rows as a matrix-vector product) vs fp64: plain, with the pre-activation affine +
    ReLU on the input, with residual and output ReLU, a padded bank (42 real of 64 columns) and 63 / 64 / 4096 outputs."""
    from hn_amd import ops
    from hn_amd.pose2mesh_engine import _dense
    gen = torch.Generator().manual_seed(41)
    for m, kx, n in ((1, 42, 4096), (3, 4096, 4096), (4, 4096, 63), (2, 4096, 64), (1, 96, 40)):
        w = torch.randn((n, kx), generator=gen, dtype=torch.float64) * (2.0 / kx) ** 0.5
        b = torch.randn((n,), generator=gen, dtype=torch.float64)
        cw = _dense(w, b, "cuda")
        x = torch.randn((m, kx), generator=gen)
        sc, sh = torch.rand((kx,), generator=gen) + 0.5, torch.randn((kx,), generator=gen) * 0.3
        res = torch.randn((m, n), generator=gen)
        y0 = ops.linear_rows(x.cuda(), cw).cpu().double()
        want0 = x.double() @ w.T + b
        y1 = ops.linear_rows(x.cuda(), cw, scale=sc.cuda(), shift=sh.cuda(), residual=res.cuda(), relu=True).cpu().double()
        want1 = torch.relu(torch.relu(x.double() * sc.double() + sh.double()) @ w.T + b + res.double())
        tol = 2e-6 * max(1.0, want0.abs().max().item()) * max(1.0, (kx / 64) ** 0.5)
        assert (y0 - want0).abs().max().item() < tol and (y1 - want1).abs().max().item() < tol, (m, kx, n)
    with pytest.raises(RuntimeError, match="1..4 rows"):
        ops.linear_rows(torch.zeros((5, 64)).cuda(), cw)

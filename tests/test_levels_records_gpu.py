"""Round-3 helpers through the C ABI on the GPU: the levels-merged GroupNorm finalize / apply passes (bit-identical to
the per-level launches they replace), the per-frame record pack / unpack kernels of the N > 1 gather (bit-identical to
the host-side record layout of hn_amd.dist), and the non-finite counter."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _levels(n, c, dims, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn((n, h, w, c), generator=g).cuda() * (1.0 + l) + 0.3 * l for l, (h, w) in enumerate(dims)]


@pytest.mark.parametrize("n,dims", [(1, [(100, 136), (50, 68), (25, 34)]), (3, [(13, 17), (7, 9), (8, 4)])])
def test_levels_finalize_and_split_equal_per_level_launches(n, dims):
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    c = 512
    xs = [ops.to_split(t) for t in _levels(n, 256, dims, seed=1)]
    g = torch.Generator().manual_seed(2)
    w = torch.randn((c, 3, 3, 256), generator=g) * 0.02
    wd, w16 = w.cuda(), split_f16x3(w).cuda()
    bias = torch.randn((c,), generator=g).cuda()
    gamma, beta = (torch.rand((c,), generator=g) + 0.5).cuda(), torch.randn((c,), generator=g).cuda()
    parts, ys = [], []
    for x in xs:   # conv epilogue writes the raw output and the GroupNorm partial sums
        hw = x.shape[1] * x.shape[2]
        part = torch.zeros((ops.gn_rows32_scratch_floats(n * hw, c),), device="cuda")
        ys.append(ops.conv2d_nhwc(x, wd, bias, pad=1, w16=w16, gn_partial=part))
        parts.append(part)
    hws = [x.shape[1] * x.shape[2] for x in xs]
    ref_aff = [ops.groupnorm_finalize_rows32(parts[l], gamma, beta, n, hws[l], 64) for l in range(len(xs))]
    aff = ops.groupnorm_finalize_rows32_levels(parts, gamma, beta, n, hws, 64)
    for (s0, t0), (s1, t1) in zip(ref_aff, aff):
        assert torch.equal(s0, s1) and torch.equal(t0, t1)
    ref_a = [ops.to_split(ys[l], *ref_aff[l], relu=True) for l in range(len(xs))]
    a = ops.to_split_levels(ys, aff, relu=True)
    for r, q in zip(ref_a, a):
        assert r.shape == q.shape and torch.equal(r, q)
    # against the definition (fp64 GroupNorm with 64 groups of 8 channels + ReLU)
    y0 = ys[0].double().cpu()
    gn = torch.nn.functional.group_norm(y0.permute(0, 3, 1, 2), 64, gamma.double().cpu(), beta.double().cpu(), 1e-5)
    want = gn.clamp_min(0).permute(0, 2, 3, 1)
    got = ops.from_split(a[0]).double().cpu()
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("n,rows", [(5, 5), (3, 8), (0, 4)])
def test_record_kernels_match_host_layout(n, rows):
    """hn_pack_records == the torch-op packing of hn_amd.dist (the gloo / CPU path), byte for byte; unpack inverts it."""
    from hn_amd import dist as hdist
    from hn_amd import ops
    g = torch.Generator().manual_seed(7)
    kp = torch.randn((n, 21, 3), generator=g)
    box = torch.randint(-5, 700, (n, 4), generator=g, dtype=torch.int64)
    has = torch.randint(0, 2, (n,), generator=g, dtype=torch.int32)
    rec_bytes = hdist._record_bytes(63)
    rec = ops.pack_records(kp.cuda(), box.cuda(), has.cuda(), rows, rec_bytes)
    hk, hb, hh, hv = hdist.gather_results(kp, box, has, per_rank=rows)          # host path, no process group
    send, _ = hdist._buffers(rows, 1, rec_bytes, kp.device)
    assert torch.equal(rec.cpu(), send)
    k2, b2, h2, v2 = ops.unpack_records(rec, 21)
    assert torch.equal(k2.cpu(), hk) and torch.equal(b2.cpu(), hb) and torch.equal(h2.cpu(), hh)
    assert torch.equal(v2.bool().cpu(), hv) and v2.sum().item() == n
    # the device path of gather_results (no process group: pack + unpack only)
    if n:
        gk, gb, gh, gv = hdist.gather_results(kp.cuda(), box.cuda(), has.cuda(), per_rank=rows)
        assert torch.equal(gk.cpu(), hk) and torch.equal(gb.cpu(), hb) and torch.equal(gh.cpu(), hh) and torch.equal(gv.cpu(), hv)


def test_nonfinite_count():
    from hn_amd import ops
    x = torch.randn((32, 21, 3)).cuda()
    assert ops.nonfinite_count(x).item() == 0
    x[3, 4, 1] = float("nan")
    x[9, 0, 0] = float("inf")
    x[31, 20, 2] = -float("inf")
    flag = ops.nonfinite_count(x)
    assert flag.item() == 3
    assert ops.nonfinite_count(x, flag).item() == 6   # accumulates into a caller-owned flag

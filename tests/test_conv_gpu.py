"""HIP implicit-GEMM conv / pool / GroupNorm vs the plain-torch CPU reference (oracle/ops_ref.py).

Tolerance: fp32 with a different accumulation order -> |err| <= 2e-5 * sqrt(K) * rms(term),
tested as rtol 1e-4 / atol 1e-4 * max|y| (values are O(1)).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _check(y_gpu, y_ref, what=""):
    y = y_gpu.cpu()
    assert y.shape == y_ref.shape, (what, y.shape, y_ref.shape)
    scale = max(1.0, float(y_ref.abs().max()))
    err = float((y - y_ref).abs().max())
    assert err <= 1e-4 * scale, f"{what}: max err {err} vs scale {scale}"


# (n, h, w, cin, cout, r, stride, pad, dil)  -- shapes taken from the A2J / FCOS tables
CASES = [
    (2, 44, 44, 64, 256, 1, 1, 0, 1),     # bottleneck 1x1 expand
    (2, 44, 44, 64, 64, 3, 1, 1, 1),      # 3x3
    (2, 44, 44, 128, 128, 3, 2, 1, 1),    # 3x3 stride 2
    (3, 11, 11, 512, 512, 3, 1, 2, 2),    # layer4 dilated 3x3, ragged M (363 rows)
    (2, 44, 44, 256, 512, 1, 2, 0, 1),    # 1x1 stride-2 downsample
    (2, 64, 48, 4, 64, 7, 2, 3, 1),       # stem, Cin padded to 4 (small-C gather)
    (2, 11, 11, 256, 336, 3, 1, 1, 1),    # head output, Cout not a tile multiple
    (1, 25, 34, 256, 5, 3, 1, 1, 1),      # FCOS head output, Cout = 5
    (1, 13, 17, 32, 48, 3, 1, 1, 1),      # odd sizes, Cin = 32
    (1, 9, 9, 8, 16, 3, 1, 1, 1),         # Cin = 8 (small-C path with 2 chunks per tap)
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 6])
def test_conv_matches_reference(case, tile):
    from hn_amd import ops
    from oracle import ops_ref
    n, h, w, cin, cout, r, stride, pad, dil = case
    x = _rand((n, h, w, cin), 1)
    wt = _rand((cout, r, r, cin), 2, scale=(2.0 / (cin * r * r)) ** 0.5)
    b = _rand((cout,), 3, 0.1)
    ref = ops_ref.conv2d_nhwc(x, wt, b, stride, pad, dil, relu_cols=cout)
    y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), b.cuda(), stride=stride, pad=pad, dil=dil, relu=True, tile=tile)
    _check(y, ref, f"case {case} tile {tile}")


def test_conv_residual_and_partial_relu():
    from hn_amd import ops
    from oracle import ops_ref
    x = _rand((2, 22, 22, 128), 4)
    wt = _rand((512, 1, 1, 128), 5, 0.1)
    res = _rand((2, 22, 22, 512), 6)
    ref = ops_ref.conv2d_nhwc(x, wt, None, relu_cols=100, residual=res)
    y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), None, relu_cols=100, residual=res.cuda())
    _check(y, ref, "residual + relu_cols")


def test_conv_fpn_upsample_add():
    from hn_amd import ops
    from oracle import ops_ref
    x = _rand((2, 50, 68, 64), 7)
    wt = _rand((256, 1, 1, 64), 8, 0.1)
    b = _rand((256,), 9, 0.1)
    top = _rand((2, 25, 34, 256), 10)
    ref = ops_ref.conv2d_nhwc(x, wt, b, residual=top, res_upsample=True)
    y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), b.cuda(), residual=top.cuda(), res_upsample=True)
    _check(y, ref, "fpn lateral + nearest 2x add")


def test_conv_groupnorm_on_load():
    """conv(relu(GN(x))) with GN folded into per-(image, channel) scale/shift applied on load."""
    from hn_amd import ops
    from oracle import ops_ref
    x = _rand((2, 25, 34, 256), 11, 2.0) + 0.3
    gamma = 1.0 + 0.2 * _rand((256,), 12)
    beta = 0.1 * _rand((256,), 13)
    wt = _rand((256, 3, 3, 256), 14, (2.0 / 2304) ** 0.5)
    b = _rand((256,), 15, 0.1)
    sc_ref, sh_ref = ops_ref.groupnorm_affine(x, gamma, beta)
    sc, sh = ops.groupnorm_affine(x.cuda(), gamma.cuda(), beta.cuda())
    _check(sc, sc_ref, "gn scale")
    _check(sh, sh_ref, "gn shift")
    xn = ops_ref.groupnorm_relu_nhwc(x, gamma, beta)
    ref = ops_ref.conv2d_nhwc(xn, wt, b, pad=1)
    y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), b.cuda(), pad=1, in_scale=sc, in_shift=sh)
    _check(y, ref, "conv with GroupNorm+ReLU on load")


def test_conv_channel_slice_views():
    from hn_amd import ops
    from oracle import ops_ref
    wide = _rand((2, 11, 11, 512), 16)
    wt = _rand((256, 3, 3, 256), 17, 0.05)
    ref = ops_ref.conv2d_nhwc(wide[..., 256:].contiguous(), wt, None, pad=1)
    wg = wide.cuda()
    out_wide = torch.zeros((2, 11, 11, 512), device="cuda")
    ops.conv2d_nhwc(wg[..., 256:], wt.cuda(), None, pad=1, out=out_wide[..., :256])
    _check(out_wide[..., :256].contiguous(), ref, "slice in / slice out")
    assert float(out_wide[..., 256:].abs().max()) == 0.0


def test_maxpool():
    from hn_amd import ops
    from oracle import ops_ref
    x = _rand((2, 89, 67, 64), 18)
    y = ops.maxpool3x3s2_nhwc(x.cuda())
    assert torch.equal(y.cpu(), ops_ref.maxpool3x3s2_nhwc(x))


def test_bad_arguments_fail_loudly():
    from hn_amd import ops
    x = torch.zeros((1, 8, 8, 6), device="cuda")  # cin not a multiple of 4
    w = torch.zeros((8, 3, 3, 6), device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.conv2d_nhwc(x, w, None, pad=1)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.conv2d_nhwc(torch.zeros((1, 8, 8, 4)), torch.zeros((8, 3, 3, 4), device="cuda"))


# ---- split-fp16 ("f16x3") convolution: fp32-grade results on the f16 MFMA ----
F16X3_CASES = [
    (2, 44, 44, 64, 256, 1, 1, 0, 1),
    (2, 44, 44, 64, 64, 3, 1, 1, 1),
    (2, 44, 44, 128, 128, 3, 2, 1, 1),
    (3, 11, 11, 512, 512, 3, 1, 2, 2),
    (2, 44, 44, 256, 512, 1, 2, 0, 1),
    (2, 11, 11, 256, 336, 3, 1, 1, 1),
    (1, 25, 34, 256, 5, 3, 1, 1, 1),
    (1, 13, 17, 32, 48, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", F16X3_CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 6, 7, 8])
def test_conv_f16x3_matches_fp64_reference(case, tile):
    """Error budget: operands carry 22 bits (hi+lo), products exact, fp32 accumulate =>
    same 1e-4*scale bar as the exact-f32 kernel, checked against an fp64 convolution."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    n, h, w, cin, cout, r, stride, pad, dil = case
    x = _rand((n, h, w, cin), 21)
    wt = _rand((cout, r, r, cin), 22, scale=(2.0 / (cin * r * r)) ** 0.5)
    b = _rand((cout,), 23, 0.1)
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), stride, pad, dil, relu_cols=cout).float()
    y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), b.cuda(), stride=stride, pad=pad, dil=dil, relu=True, tile=tile,
                        w16=split_f16x3(wt).cuda())
    _check(y, ref, f"f16x3 case {case} tile {tile}")
    # and it must be fp32-grade, not fp16-grade: two orders below the fp16 rounding level
    scale = max(1.0, float(ref.abs().max()))
    assert float((y.cpu() - ref).abs().max()) <= 2e-5 * scale


def test_conv_f16x3_small_magnitudes_keep_precision():
    """lo parts of O(1e-2) activations / weights are fp16 SUBNORMALS (~1e-5 < 6.1e-5): they must
    not be flushed by the MFMA (relative error stays ~3e-6; a flush would give ~5e-4)."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    x = _rand((2, 22, 22, 128), 31, 2e-2)
    wt = _rand((128, 3, 3, 128), 32, 1e-2)
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), None, 1, 1, 1).float()
    y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), None, pad=1, w16=split_f16x3(wt).cuda()).cpu()
    rel = float((y - ref).abs().max() / ref.abs().max())
    assert rel <= 2e-5, rel


def test_conv_f16x3_fused_paths():
    """GroupNorm-on-load + channel-slice views + FPN upsample-add, as the FCOS engine uses them."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    wide = _rand((2, 25, 34, 512), 41, 2.0) + 0.3
    gamma = 1.0 + 0.2 * _rand((512,), 42)
    beta = 0.1 * _rand((512,), 43)
    wt = _rand((256, 3, 3, 256), 44, (2.0 / 2304) ** 0.5)
    b = _rand((256,), 45, 0.1)
    sc, sh = ops.groupnorm_affine(wide.cuda(), gamma.cuda(), beta.cuda(), groups=64)
    xs = wide[..., 256:].contiguous()
    xn = ops_ref.groupnorm_relu_nhwc(xs, gamma[256:], beta[256:])
    ref = ops_ref.conv2d_nhwc(xn, wt, b, pad=1)
    y = ops.conv2d_nhwc(wide.cuda()[..., 256:], wt.cuda(), b.cuda(), pad=1, in_scale=sc[:, 256:], in_shift=sh[:, 256:],
                        w16=split_f16x3(wt).cuda())
    _check(y, ref, "f16x3 GN-on-load over a channel slice")
    x2 = _rand((2, 50, 68, 64), 46)
    w2 = _rand((256, 1, 1, 64), 47, 0.1)
    top = _rand((2, 25, 34, 256), 48)
    ref2 = ops_ref.conv2d_nhwc(x2, w2, b, residual=top, res_upsample=True)
    y2 = ops.conv2d_nhwc(x2.cuda(), w2.cuda(), b.cuda(), residual=top.cuda(), res_upsample=True,
                         w16=split_f16x3(w2).cuda())
    _check(y2, ref2, "f16x3 fpn lateral")


# ---- S32 split activation format (producers / consumers of the f16x3 convolutions) ----
def test_split_roundtrip_and_affine():
    from hn_amd import ops
    x = _rand((2, 9, 7, 64), 51, 3.0)
    xs = ops.to_split(x.cuda())
    assert ops.is_split(xs) and tuple(xs.shape) == (2, 9, 7, 2, 2, 32)
    back = ops.from_split(xs).cpu()
    # hi + lo carries 22 bits: |error| <= 2^-21 |x| (+ half an fp16 subnormal step for tiny lo parts)
    assert bool(((back - x).abs() <= 2 ** -21 * x.abs() + 3.0e-8).all())
    hi = xs[..., 0, :].reshape(2, 9, 7, 64).float().cpu()
    assert torch.equal(hi, x.half().float())                      # hi plane = fp16(x), channel order kept
    sc, sh = _rand((2, 64), 52) * 0.5 + 1.0, _rand((2, 64), 53, 0.2)
    ya = ops.from_split(ops.to_split(x.cuda(), sc.cuda(), sh.cuda(), relu=True)).cpu()
    ref = torch.relu(x * sc[:, None, None, :] + sh[:, None, None, :])
    assert float((ya - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # channel-slice input (fp32 slice of a wider tensor, sliced affine table)
    wide = _rand((2, 5, 5, 128), 54)
    scw, shw = _rand((2, 128), 55) + 2.0, _rand((2, 128), 56)
    ys = ops.from_split(ops.to_split(wide.cuda()[..., 64:], scw.cuda()[:, 64:], shw.cuda()[:, 64:], relu=False)).cpu()
    refs = wide[..., 64:] * scw[:, None, None, 64:] + shw[:, None, None, 64:]
    assert float((ys - refs).abs().max()) <= 2e-6 * float(refs.abs().max())


def test_maxpool_split_is_exact():
    from hn_amd import ops
    from oracle import ops_ref
    x = _rand((2, 45, 33, 64), 57)
    xs = ops.to_split(x.cuda())
    y = ops.from_split(ops.maxpool3x3s2_nhwc(xs)).cpu()
    assert torch.equal(y, ops_ref.maxpool3x3s2_nhwc(ops.from_split(xs).cpu()))


def test_conv_f16x3_split_in_out_residual():
    """Bottleneck tail as the engines run it: S32 input, S32 residual, ReLU, S32 output, channel slices."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    x = _rand((2, 22, 22, 256), 58)
    wt = _rand((512, 1, 1, 128), 59, 0.1)
    b = _rand((512,), 60, 0.1)
    res = _rand((2, 22, 22, 512), 61)
    xs = ops.to_split(x.cuda())                 # 8 blocks; the conv reads blocks 4..7 = channels 128..255
    rs = ops.to_split(res.cuda())
    y = ops.conv2d_nhwc(xs[:, :, :, 4:], wt.cuda(), b.cuda(), relu=True, residual=rs, w16=split_f16x3(wt).cuda(),
                        out_split=True)
    assert ops.is_split(y)
    # reference on the values the kernel actually sees (hi + lo of x and of the residual)
    ref = ops_ref.conv2d_nhwc(ops.from_split(xs).cpu()[..., 128:].contiguous(), wt, b, relu_cols=512,
                              residual=ops.from_split(rs).cpu())
    _check(ops.from_split(y), ref, "S32 in / S32 residual / S32 out")
    # fp32 residual + fp32 output into a channel slice of a wider tensor
    wide_out = torch.zeros((2, 22, 22, 1024), device="cuda")
    ops.conv2d_nhwc(xs[:, :, :, 4:], wt.cuda(), b.cuda(), residual=res.cuda(), w16=split_f16x3(wt).cuda(),
                    out=wide_out[..., 512:])
    ref2 = ops_ref.conv2d_nhwc(ops.from_split(xs).cpu()[..., 128:].contiguous(), wt, b, residual=res)
    _check(wide_out[..., 512:].contiguous(), ref2, "fp32 residual / sliced fp32 out")
    assert float(wide_out[..., :512].abs().max()) == 0.0


@pytest.mark.parametrize("n,h,w,cin,cout,groups,tile", [
    (3, 10, 13, 64, 256, 32, 0),     # hw = 130: row groups straddle image boundaries, M not a multiple of 32
    (2, 25, 34, 256, 512, 64, 1),    # tower0 shape of the P5 level (two stacked GroupNorm(32,256))
    (2, 9, 7, 32, 64, 8, 2),         # 128x64 tile, hw = 63
    (5, 6, 6, 32, 64, 4, 3),         # hw = 36: a 32-row group can end in the next image; 16 channels per group
    (2, 20, 20, 64, 64, 8, 8),       # 256x64 tile
])
def test_conv_f16x3_groupnorm_partials_in_epilogue(n, h, w, cin, cout, groups, tile):
    """GroupNorm statistics emitted by the conv epilogue + rows32 finalize == the stand-alone statistics pass
    over the same output (both fp32 partials combined in fp64), and the conv output itself is unchanged."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(n * 1000 + cout)
    x = torch.randn((n, h, w, cin), generator=g)
    wt = torch.randn((cout, 3, 3, cin), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g)
    gamma = torch.rand((cout,), generator=g) + 0.5
    beta = torch.randn((cout,), generator=g) * 0.1
    x16 = ops.to_split(x.cuda())
    w16 = split_f16x3(wt).cuda()
    part = torch.full((ops.gn_rows32_scratch_floats(n * h * w, cout),), float("nan"), device="cuda")
    y = ops.conv2d_nhwc(x16, wt.cuda(), b.cuda(), pad=1, w16=w16, tile=tile, gn_partial=part)
    y_plain = ops.conv2d_nhwc(x16, wt.cuda(), b.cuda(), pad=1, w16=w16, tile=tile, splitk=False)
    assert torch.equal(y, y_plain)
    assert not torch.isnan(part).any()          # every (row group, channel unit) record is written exactly once
    sc, sh = ops.groupnorm_finalize_rows32(part, gamma.cuda(), beta.cuda(), n, h * w, groups)
    sc_ref, sh_ref = ops.groupnorm_affine(y, gamma.cuda(), beta.cuda(), groups=groups)
    assert (sc - sc_ref).abs().max().item() <= 2e-6 * sc_ref.abs().max().item()
    assert (sh - sh_ref).abs().max().item() <= 2e-6 * max(1.0, sh_ref.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(x16, wt.cuda(), b.cuda(), pad=1, w16=w16, relu=True, gn_partial=part)


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, r, stride, dil, residual ("s32" / "up" / None), out_split
    (1, 11, 11, 1024, 256, 3, 1, 1, None, True),      # 8 workgroups x 288 k tiles -> 16 splits
    (2, 11, 11, 2048, 512, 3, 1, 1, "s32", True),     # A2J layer4-style bottleneck conv with a split residual
    (1, 25, 34, 4096, 256, 1, 1, 1, "up", False),     # 1x1 + nearest-2x top-down add (FPN style), fp32 out
    (3, 11, 11, 512, 512, 3, 1, 2, None, True),       # dilated, 144 k tiles
    (1, 7, 5, 512, 336, 3, 1, 1, None, False),        # ragged M (35 rows), Cout not a multiple of the tile
])
def test_conv_f16x3_splitk_matches_plain_and_reference(case):
    """Small grids run split-K (partial tiles in a workspace, summed in fixed order by a second launch): same
    fp32-grade error bar as the single-pass kernel, bitwise repeatable, and every epilogue mode still applies."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    if not ops.SPLITK:
        pytest.skip("split-K disabled by HN_SPLITK=0")
    n, h, w, cin, cout, r, stride, dil, res, osplit = case
    pad = dil * (r // 2)
    x = _rand((n, h, w, cin), 41)
    wt = _rand((cout, r, r, cin), 42, scale=(2.0 / (cin * r * r)) ** 0.5)
    b = _rand((cout,), 43, 0.1)
    oh, ow = ops.conv_out_size(h, w, r, r, stride, pad, dil)
    kw = dict(stride=stride, pad=pad, dil=dil, relu=True, w16=split_f16x3(wt).cuda(), out_split=osplit)
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), stride, pad, dil, relu_cols=0)
    if res == "s32":
        rt = _rand((n, oh, ow, cout), 44)
        kw["residual"] = ops.to_split(rt.cuda())
        ref = ref + rt.double()
    elif res == "up":
        rt = _rand((n, (oh + 1) // 2, (ow + 1) // 2, cout), 44)
        kw["residual"], kw["res_upsample"] = rt.cuda(), True
        iy = (torch.arange(oh) * rt.shape[1]) // oh
        ix = (torch.arange(ow) * rt.shape[2]) // ow
        ref = ref + rt.double()[:, iy][:, :, ix]
    ref = torch.relu(ref).float()
    x16 = ops.to_split(x.cuda())
    y_split = ops.conv2d_nhwc(x16, wt.cuda(), b.cuda(), **kw)
    y_again = ops.conv2d_nhwc(x16, wt.cuda(), b.cuda(), **kw)
    y_plain = ops.conv2d_nhwc(x16, wt.cuda(), b.cuda(), splitk=False, **kw)
    assert torch.equal(y_split, y_again)
    f = (lambda t: ops.from_split(t)) if osplit else (lambda t: t)
    scale = max(1.0, float(ref.abs().max()))
    assert float((f(y_split).cpu() - ref).abs().max()) <= 2e-5 * scale
    assert float((f(y_plain).cpu() - ref).abs().max()) <= 2e-5 * scale
    assert not torch.equal(y_split, y_plain)   # the split path really ran (different summation order)


def test_conv_f16x3_random_shapes_auto_tile():
    """Seeded sweep over ragged shapes through the AUTO tile / split-K heuristics (whatever they pick must be right):
    odd spatial sizes, Cout not a multiple of any tile, strides, dilation, 1x1 and 3x3, S32 and fp32 outputs."""
    import random
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    rnd = random.Random(1234)
    for case in range(24):
        n, h, w = rnd.randint(1, 3), rnd.randint(1, 40), rnd.randint(1, 40)
        cin = rnd.choice([32, 64, 96, 160, 512])
        cout = rnd.choice([8, 24, 40, 64, 72, 136, 256, 5, 3])
        r = rnd.choice([1, 3])
        stride, dil = rnd.choice([1, 1, 2]), rnd.choice([1, 1, 2])
        pad = dil * (r // 2)
        if (h + 2 * pad - dil * (r - 1) - 1) < 0 or (w + 2 * pad - dil * (r - 1) - 1) < 0:
            continue
        x = _rand((n, h, w, cin), 100 + case)
        wt = _rand((cout, r, r, cin), 200 + case, scale=(2.0 / (cin * r * r)) ** 0.5)
        b = _rand((cout,), 300 + case, 0.1)
        osplit = cout % 32 == 0 and rnd.random() < 0.5
        ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), stride, pad, dil, relu_cols=cout).float()
        y = ops.conv2d_nhwc(ops.to_split(x.cuda()), wt.cuda(), b.cuda(), stride=stride, pad=pad, dil=dil, relu=True,
                            w16=split_f16x3(wt).cuda(), out_split=osplit)
        y = ops.from_split(y) if osplit else y
        scale = max(1.0, float(ref.abs().max()))
        err = float((y.cpu() - ref).abs().max())
        assert y.shape == ref.shape and err <= 2e-5 * scale, (case, (n, h, w, cin, cout, r, stride, dil), err)


def test_conv_f16x3_grouped_equals_separate_launches():
    """hn_conv2d_nhwc_f16x3_grouped: three same-shape convolutions in one launch (gridDim.z) give bit-identical
    results to three launches -- S32 outputs with ReLU, fp32 outputs with GroupNorm partial sums, slice inputs."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(77)
    n, h, w, cin, cout = 2, 11, 11, 256, 256
    wide = ops.to_split(torch.randn((n, h, w, 3 * cin), generator=g).cuda())
    xs = [wide[:, :, :, 8 * i:8 * (i + 1)] for i in range(3)]          # channel-slice views, equal pixel stride
    ws = []
    for i in range(3):
        wt = torch.randn((cout, 3, 3, cin), generator=g) * (2.0 / (9 * cin)) ** 0.5
        ws.append(NS(w=wt.cuda(), bias=torch.randn((cout,), generator=g).cuda(), w16=split_f16x3(wt).cuda()))
    ys = ops.conv2d_nhwc_grouped(xs, ws, pad=1, relu=True, out_split=True)
    for x, cw, y in zip(xs, ws, ys):
        ref = ops.conv2d_nhwc(x, cw.w, cw.bias, pad=1, relu=True, w16=cw.w16, out_split=True, splitk=False)
        assert torch.equal(y, ref)
    need = ops.gn_rows32_scratch_floats(n * h * w, cout)
    parts = [torch.full((need,), float("nan"), device="cuda") for _ in range(2)]
    ys = ops.conv2d_nhwc_grouped(xs[:2], ws[:2], pad=1, gn_partials=parts)
    for x, cw, y, part in zip(xs, ws, ys, parts):
        ref_part = torch.empty_like(part)
        ref = ops.conv2d_nhwc(x, cw.w, cw.bias, pad=1, w16=cw.w16, gn_partial=ref_part)
        assert torch.equal(y, ref) and torch.equal(part, ref_part)
    with pytest.raises(ValueError):
        ops.conv2d_nhwc_grouped([xs[0], wide[:, :, :, :16]], ws[:2], pad=1)     # different shapes


def test_conv_f16x3_grouped_stacked_outputs_and_shared_gn_slab():
    """Two grouped members writing channel halves of ONE fp32 tensor and ONE GroupNorm slab (the FCOS towers):
    equals the two separate convolutions, and the 64-group finalize of the shared slab equals the statistics pass."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(78)
    n, h, w, c = 3, 10, 13, 256
    a = ops.to_split(torch.randn((n, h, w, 2 * c), generator=g).cuda())
    xs = [a[:, :, :, :8], a[:, :, :, 8:]]
    ws = []
    for i in range(2):
        wt = torch.randn((c, 3, 3, c), generator=g) * (2.0 / (9 * c)) ** 0.5
        ws.append(NS(w=wt.cuda(), bias=torch.randn((c,), generator=g).cuda(), w16=split_f16x3(wt).cuda()))
    part = torch.full((ops.gn_rows32_scratch_floats(n * h * w, 2 * c),), float("nan"), device="cuda")
    y = torch.empty((n, h, w, 2 * c), device="cuda")
    ops.conv2d_nhwc_grouped(xs, ws, pad=1, outs=[y, y], out_channel_offsets=[0, c], gn=[(part, 0), (part, c // 8)],
                            gn_units=2 * c // 8)
    assert not torch.isnan(part).any()
    for i in range(2):
        ref = ops.conv2d_nhwc(xs[i], ws[i].w, ws[i].bias, pad=1, w16=ws[i].w16, splitk=False)
        assert torch.equal(y[..., i * c:(i + 1) * c], ref)
    gamma = (torch.rand((2 * c,), generator=g) + 0.5).cuda()
    beta = (torch.randn((2 * c,), generator=g) * 0.1).cuda()
    sc, sh = ops.groupnorm_finalize_rows32(part, gamma, beta, n, h * w, 64)
    sc_ref, sh_ref = ops.groupnorm_affine(y, gamma, beta, groups=64)
    assert (sc - sc_ref).abs().max().item() <= 2e-6 * sc_ref.abs().max().item()
    assert (sh - sh_ref).abs().max().item() <= 2e-6 * max(1.0, sh_ref.abs().max().item())


def test_conv_f16x3_grouped_members_of_different_spatial_size():
    """The FPN-level case: same weights / channels / batch, three map sizes, one launch; each member equals its own
    single launch, including the 5-channel (scalar-epilogue) output convs with a partial ReLU."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(79)
    n, cin = 2, 64
    xs = [ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda()) for h, w in ((20, 28), (10, 14), (5, 7))]
    for cout, rc in ((128, None), (5, 4)):
        wt = torch.randn((cout, 3, 3, cin), generator=g) * (2.0 / (9 * cin)) ** 0.5
        cw = NS(w=wt.cuda(), bias=torch.randn((cout,), generator=g).cuda(), w16=split_f16x3(wt).cuda())
        ys = ops.conv2d_nhwc_grouped(xs, [cw] * 3, pad=1, relu_cols=rc)
        for x, y in zip(xs, ys):
            ref = ops.conv2d_nhwc(x, cw.w, cw.bias, pad=1, relu_cols=rc, w16=cw.w16, splitk=False)
            assert y.shape == ref.shape and torch.equal(y, ref)


# ---- f16x3 range contract (include/handnet_hip.h: hn_range_check_enable / _fetch) ----
def test_f16x3_range_contract_flags_overflow_instead_of_returning_inf():
    """|v| > 65504 cannot be split (hi = fp16(v) = inf).  With the debug switch on, every split producer raises
    the sticky flag: the fp32 -> S32 pass, and a conv epilogue that writes S32."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    ops.range_check_enable(True)
    try:
        assert ops.range_check_fetch(reset=True) in (False, True)      # clear whatever earlier tests left
        x = _rand((1, 9, 9, 64), 71, 1.0).cuda()
        ops.to_split(x)
        assert ops.range_check_fetch() is False
        big = x.clone()
        big[0, 4, 4, 7] = 1.0e5
        xs = ops.to_split(big)
        assert ops.range_check_fetch() is True                          # flagged ...
        assert not torch.isfinite(ops.from_split(xs)).all()             # ... because the data really is inf
        assert ops.range_check_fetch() is False                         # sticky flag was reset by the fetch
        nan = x.clone()
        nan[0, 0, 0, 0] = float("nan")
        ops.to_split(nan)
        assert ops.range_check_fetch() is True
        # conv epilogue: in-range operands, out-of-range S32 OUTPUT (64 channels x 9 taps x 40 x 40 = 9e5)
        wt = torch.full((64, 3, 3, 64), 40.0)
        xin = ops.to_split(torch.full((1, 9, 9, 64), 40.0, device="cuda"))
        assert ops.range_check_fetch() is False
        y32 = ops.conv2d_nhwc(xin, wt.cuda(), None, pad=1, w16=split_f16x3(wt).cuda(), out_split=False)
        assert ops.range_check_fetch() is False and torch.isfinite(y32).all()   # fp32 output is fine
        ops.conv2d_nhwc(xin, wt.cuda(), None, pad=1, w16=split_f16x3(wt).cuda(), out_split=True)
        assert ops.range_check_fetch() is True
    finally:
        ops.range_check_enable(False)
    ops.to_split(big)                                                    # switch off: nothing is recorded
    ops.range_check_enable(True)
    try:
        assert ops.range_check_fetch() is False
    finally:
        ops.range_check_enable(False)


def test_nan_activations_are_flagged_and_stay_nan_behind_a_relu():
    """a NaN that is BORN inside the network (here: an fp32 residual that carries one) must not be laundered to 0 by
    the ReLU of the epilogue that meets it -- torch.relu(NaN) is NaN -- and the range note must see it: the magnitude maximum is
    NaN-propagating (v_maximum3_f32), so word 0 of the flag block is set.  Also through the S32 max pooling."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    x = _rand((1, 12, 12, 64), 81, 1.0).cuda()
    wt = _rand((64, 3, 3, 64), 82, 0.05)
    res = _rand((1, 12, 12, 64), 83, 1.0).cuda()
    res[0, 5, 6, 9] = float("nan")
    block = torch.zeros((4,), device="cuda", dtype=torch.int32)
    with ops.range_scope(block):
        xs = ops.to_split(x)
        assert ops.range_check_collect(block).cpu().tolist() == [0, 0, 0, 0]
        ys = ops.conv2d_nhwc(xs, wt.cuda(), None, pad=1, relu=True, residual=res, w16=split_f16x3(wt).cuda(), out_split=True)
        words = ops.range_check_collect(block).cpu().tolist()
        y = ops.from_split(ys)
        assert words[0] == 1, words
        assert torch.isnan(y[0, 5, 6, 9]) and int(torch.isnan(y).sum()) == 1          # NaN, not relu -> 0
        pooled = ops.from_split(ops.maxpool3x3s2_nhwc(ys))
        assert int(torch.isnan(pooled[..., 9]).sum()) >= 1 and torch.isfinite(pooled[..., :9]).all()
    clean = ops.conv2d_nhwc(xs, wt.cuda(), None, pad=1, relu=True, residual=torch.nan_to_num(res), w16=split_f16x3(wt).cuda(),
                            out_split=True)
    ok = torch.ones_like(y, dtype=torch.bool)
    ok[0, 5, 6, 9] = False
    assert torch.equal(y[ok], ops.from_split(clean)[ok])


def test_range_scopes_of_two_host_threads_do_not_mix():
    """the switch and the bound flag block are state of the calling HOST THREAD.  While thread A holds a
    scope on its block, thread B opens its own, overflows, and closes it: B's block is flagged, A's is not, A's scope is still
    in force afterwards, and a thread without a scope notes nowhere."""
    import threading
    from hn_amd import ops
    big = torch.full((1, 4, 4, 32), 1.0e5, device="cuda")
    a_block = torch.zeros((4,), device="cuda", dtype=torch.int32)
    b_block = torch.zeros((4,), device="cuda", dtype=torch.int32)
    a_in, b_done = threading.Event(), threading.Event()
    seen = {}

    def thread_a():
        with ops.range_scope(a_block):
            a_in.set()
            b_done.wait(30)
            seen["a_enabled_after_b"] = ops._lib.load().hn_range_check_enabled()
            ops.to_split(torch.ones((1, 4, 4, 32), device="cuda"))
            seen["a_words_clean"] = ops.range_check_collect(a_block).cpu().tolist()
            ops.to_split(big)
            seen["a_words"] = ops.range_check_collect(a_block).cpu().tolist()
        seen["a_enabled_outside"] = ops._lib.load().hn_range_check_enabled()

    def thread_b():
        a_in.wait(30)
        seen["b_enabled_before"] = ops._lib.load().hn_range_check_enabled()
        ops.to_split(big)                                  # no scope on this thread: noted nowhere
        with ops.range_scope(b_block):
            ops.to_split(big)
            seen["b_words"] = ops.range_check_collect(b_block).cpu().tolist()
        with ops.range_scope(b_block, on=False):           # an engine with noting off does not switch A off
            ops.to_split(big)
        seen["b_words_off"] = ops.range_check_collect(b_block).cpu().tolist()
        b_done.set()

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start(); ta.join(60); tb.join(60)
    assert seen == {"b_enabled_before": 0, "b_words": [1, 0, 0, 0], "b_words_off": [0, 0, 0, 0], "a_enabled_after_b": 1,
                    "a_words_clean": [0, 0, 0, 0], "a_words": [1, 0, 0, 0], "a_enabled_outside": 0}, seen


def test_f16x3_wide_dynamic_range_keeps_precision():
    """Log-uniform magnitudes over nine decades (1e-6 .. 1e3) in activations and 1e-4 .. 1e1 in weights: error
    <= 2e-5 of the output scale against fp64 (the stored value's error is max(2^-22 |v|, 2^-25): tiny values lose
    relative precision but cannot hurt an output dominated by the large ones)."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    g = torch.Generator().manual_seed(72)
    mag = 10.0 ** (torch.rand((2, 20, 20, 128), generator=g) * 9.0 - 6.0)
    x = mag * torch.sign(torch.randn((2, 20, 20, 128), generator=g))
    wmag = 10.0 ** (torch.rand((64, 3, 3, 128), generator=g) * 5.0 - 4.0)
    wt = wmag * torch.sign(torch.randn((64, 3, 3, 128), generator=g))
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), None, 1, 1, 1)
    ops.range_check_enable(True)
    try:
        ops.range_check_fetch()
        y = ops.conv2d_nhwc(x.cuda(), wt.cuda(), None, pad=1, w16=split_f16x3(wt).cuda()).cpu()
        assert ops.range_check_fetch() is False
    finally:
        ops.range_check_enable(False)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


_FORMS = ("conv_no_rs", "conv_no_rs32", "split_generic", "conv_no_halo", "preprocess_generic", "conv_no_multi",
          "no_fuse_last_gn", "no_thin_outputs", "thin_form_tap", "thin_form_flat", "splitk_fill512", "conv_no_stream", "conv_no_mixed", "conv_no_deepk", "conv_no_fused_reduce")


@pytest.fixture(autouse=True)
def _kernel_forms_are_reset():
    """Tests select older kernel forms with ops.set_form (the library never reads the environment); every form is off again
    when a test ends."""
    yield
    import torch as _t
    if _t.cuda.is_available():
        from hn_amd import ops
        for f in _FORMS:
            ops.set_form(f, False)


# ---- row-shared A operand (RS kernels): the A tile of a filter ROW is staged once, taps read it at slot offsets ----
RS_CASES = [
    # n, h, w, cin, cout: image rows shorter / longer than a tile, widths that put row ends at every slot phase,
    # M not a multiple of the tile, one-row and one-column images, the tower / body shapes at small batch
    (2, 100, 136, 64, 128), (1, 25, 34, 64, 128), (3, 13, 17, 32, 128), (2, 7, 131, 32, 64), (1, 9, 129, 64, 128),
    (2, 64, 5, 32, 128), (1, 1, 300, 32, 128), (1, 300, 1, 32, 64), (4, 11, 11, 96, 128), (1, 50, 68, 256, 256),
    (1, 3, 6, 32, 128),
]


@pytest.mark.parametrize("case", RS_CASES)
@pytest.mark.parametrize("tile", [1, 2, 4])
def test_conv_f16x3_row_shared_a_is_bit_identical_to_per_tap_form(case, tile, monkeypatch):
    """Same k order and same operands => the row-shared kernel must reproduce the per-tap kernel bit for bit (S32 and fp32
    outputs, residual + ReLU), and both are fp32-grade against an fp64 convolution.  ops.set_form("conv_no_rs") selects the per-tap
    form; cases whose rows are too short for the gap slots of a wide tile fall back to it on their own."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    n, h, w, cin, cout = case
    x = _rand((n, h, w, cin), 31)
    wt = _rand((cout, 3, 3, cin), 32, scale=(2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 33, 0.1)
    res = _rand((n, h, w, cout), 34)
    xs, w16, rs32 = ops.to_split(x.cuda()), split_f16x3(wt).cuda(), ops.to_split(res.cuda())
    outs = {}
    for mode in ("rs", "per_tap"):
        ops.set_form("conv_no_rs", mode == "per_tap")
        y32 = ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), pad=1, relu=True, tile=tile, w16=w16, splitk=False)
        y16 = ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), pad=1, relu=True, residual=rs32, tile=tile, w16=w16, out_split=True,
                              splitk=False)
        outs[mode] = (y32, ops.from_split(y16))
    assert torch.equal(outs["rs"][0], outs["per_tap"][0]), f"fp32 output differs, case {case} tile {tile}"
    assert torch.equal(outs["rs"][1], outs["per_tap"][1]), f"S32 output differs, case {case} tile {tile}"
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), 1, 1, 1, relu_cols=cout).float()
    assert float((outs["rs"][0].cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_conv_f16x3_row_shared_a_grouped_levels_and_gn_partials(monkeypatch):
    """Grouped launch over FPN-like levels (three map sizes, one of them with rows shorter than the tile) with GroupNorm
    partial sums in the epilogue: outputs and partial slabs equal the per-tap form bit for bit."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(83)
    n, cin, cout = 2, 64, 128
    xs = [ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda()) for h, w in ((40, 56), (20, 28), (10, 14))]
    wt = torch.randn((cout, 3, 3, cin), generator=g) * (2.0 / (9 * cin)) ** 0.5
    cw = NS(w=wt.cuda(), bias=torch.randn((cout,), generator=g).cuda(), w16=split_f16x3(wt).cuda())
    got = {}
    for mode in ("rs", "per_tap"):
        ops.set_form("conv_no_rs", mode == "per_tap")
        parts = [torch.zeros(ops.gn_rows32_scratch_floats(x.shape[0] * x.shape[1] * x.shape[2], cout), device="cuda") for x in xs]
        ys = ops.conv2d_nhwc_grouped(xs, [cw] * 3, pad=1, gn_partials=parts)
        got[mode] = (ys, parts)
    for a, b in zip(got["rs"][0], got["per_tap"][0]):
        assert torch.equal(a, b)
    for a, b in zip(got["rs"][1], got["per_tap"][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(2, 25, 34, 64, 5), (1, 100, 136, 32, 5), (3, 13, 17, 32, 3)])
def test_conv_f16x3_row_shared_a_few_output_channels(case, monkeypatch):
    """The FCOS output convs (5 / 3 channels, scalar epilogue, 128x32 tile): row-shared == per-tap bit for bit, with a
    partial ReLU as the regression + centerness head has it."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    n, h, w, cin, cout = case
    xs = ops.to_split(_rand((n, h, w, cin), 41).cuda())
    wt = _rand((cout, 3, 3, cin), 42, scale=(2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 43, 0.1).cuda()
    w16 = split_f16x3(wt).cuda()
    got = []
    for no_rs in (False, True):
        ops.set_form("conv_no_rs", no_rs)
        got.append(ops.conv2d_nhwc(xs, wt.cuda(), b, pad=1, relu_cols=cout - 1, w16=w16, splitk=False))
    assert torch.equal(got[0], got[1])


@pytest.mark.parametrize("shape", [(2, 9, 7, 64), (3, 25, 34, 512), (1, 1, 1, 32), (2, 37, 3, 256), (1, 50, 68, 2048), (2, 5, 5, 96)])
def test_affine_split_group_stationary_kernel_equals_generic(shape, monkeypatch):
    """hn_affine_split_f32 picks a channel-group-stationary kernel for power-of-two channel counts (one image per
    blockIdx.y, scale / shift loaded once per lane, several pixels in flight): same bits as the generic kernel
    (form "split_generic"), with and without the affine, odd pixel counts, and a sliced (strided) input; C = 96 takes the
    generic kernel on both sides."""
    from hn_amd import ops
    n, h, w, c = shape
    x = _rand(shape, 61, 2.0).cuda()
    sc, sh = (_rand((n, c), 62) * 0.5 + 1.0).cuda(), _rand((n, c), 63, 0.2).cuda()
    wide = _rand((n, h, w, 2 * c), 64).cuda()
    outs = []
    for generic in (False, True):
        ops.set_form("split_generic", generic)
        outs.append((ops.to_split(x), ops.to_split(x, sc, sh, relu=True), ops.to_split(x, sc, sh, relu=False),
                     ops.to_split(wide[..., c:], sc, sh, relu=True)))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = torch.relu(x * sc[:, None, None, :] + sh[:, None, None, :]).cpu()
    got = ops.from_split(outs[0][1]).cpu()
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("n,h,w,cin,res", [(3, 200, 272, 64, True), (3, 200, 272, 64, False), (60, 44, 44, 64, False),
                                            (9, 100, 135, 128, True), (20, 67, 93, 32, False)])
def test_halo_patch_kernel_is_bit_identical_to_the_implicit_gemm(n, h, w, cin, res):
    """conv3x3_halo_kernel (3x3 / stride 1 / pad 1, 64 output channels, >= 512 tiles of 16 x 16 pixels: ResNet-34 layer1)
    against the implicit-GEMM kernel it replaces for that shape (form "conv_no_halo"): same k order, same bits -- on maps that
    are not multiples of the tile (partial tiles right / bottom), 1-4 channel blocks, with and without the S32 residual --
    and against an fp64 convolution."""
    import os
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    x = _rand((n, h, w, cin), 51)
    wt = _rand((64, 3, 3, cin), 52, scale=(2.0 / (cin * 9)) ** 0.5)
    b = _rand((64,), 53, 0.1)
    r = _rand((n, h, w, 64), 54) if res else None
    xs, w16 = ops.to_split(x.cuda()), split_f16x3(wt).cuda()
    rs = ops.to_split(r.cuda()) if res else None
    d = ops.make_conv_desc(n, h, w, cin, 64, 3, 3, 1, 1, 1, 64, 1 if res else 0)
    d.out_split, d.res_split = 1, 1 if res else 0
    assert ops._lib.load().hn_conv2d_f16x3_uses_halo(d, 1 if res else 0) == 1
    kw = dict(pad=1, relu=True, w16=w16, out_split=True, residual=rs)
    y_halo = ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), **kw)
    ops.set_form("conv_no_halo", True)
    try:
        assert ops._lib.load().hn_conv2d_f16x3_uses_halo(d, 1 if res else 0) == 0
        y_gemm = ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), **kw)
    finally:
        ops.set_form("conv_no_halo", False)
    assert torch.equal(y_halo, y_gemm)
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), 1, 1, 1, relu_cols=0)
    ref = torch.relu(ref + (r.double() if res else 0.0)).float()
    got = ops.from_split(y_halo).cpu()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_f16x1_throughput_mode_is_plain_fp16_arithmetic():
    """hn_conv_desc.terms = 1 (engines: precision="f16x1"; SURVEY D6's throughput mode, reported BESIDE the headline): only
    the hi*hi term is issued, i.e. the result is that of fp16-rounded operands with fp32 accumulation -- checked against
    exactly that reference -- and it is two to three orders less accurate than the default three-term mode."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    for (n, h, w, cin, cout, r) in [(2, 44, 44, 256, 256, 3), (1, 100, 136, 128, 128, 3), (2, 22, 22, 512, 128, 1)]:
        x = _rand((n, h, w, cin), 61)
        wt = _rand((cout, r, r, cin), 62, scale=(2.0 / (cin * r * r)) ** 0.5)
        b = _rand((cout,), 63, 0.1)
        exact = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), 1, r // 2, 1, relu_cols=cout).float()
        half = ops_ref.conv2d_nhwc(x.half().double(), wt.half().double(), b.double(), 1, r // 2, 1, relu_cols=cout).float()
        kw = dict(pad=r // 2, relu=True, w16=split_f16x3(wt).cuda())
        y3 = ops.conv2d_nhwc(x.cuda(), wt.cuda(), b.cuda(), **kw).cpu()
        with ops.f16_terms(1):
            y1 = ops.conv2d_nhwc(x.cuda(), wt.cuda(), b.cuda(), **kw).cpu()
        scale = max(1.0, float(exact.abs().max()))
        assert float((y1 - half).abs().max()) <= 2e-5 * scale          # it IS the fp16-operand convolution
        e3, e1 = float((y3 - exact).abs().max()), float((y1 - exact).abs().max())
        assert e3 <= 2e-5 * scale and e1 > 30 * e3, (e3, e1)           # and far from fp32 grade


@pytest.mark.parametrize("n,cout,relu_cols,sizes", [
    (2, 5, 0, [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)]),     # the FCOS pyramid of an 800x1088 frame
    (1, 5, 4, [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)]),
    (3, 8, 3, [(33, 47), (16, 16), (1, 1)]),
    (1, 16, 16, [(17, 15)]),
    (2, 1, 0, [(5, 40), (40, 5)]),
    (8, 5, 4, [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)]),    # > 256 flat-pixel ranges: unit_len above one step
    (1, 5, 0, [(3, 167)]),                                               # the widest level the LDS ring takes
    (1, 4, 2, [(1, 2), (1, 300), (300, 1), (2, 2)]),                     # 300 wide: falls back to the tap kernel
    (3, 5, 4, [(1, 2), (2, 1), (2, 3), (7, 9), (13, 17)]),              # levels smaller than one step / than the lead-in
    (1, 3, 1, [(1, 167), (167, 1)]),                                     # one-row and one-column maps
])
@pytest.mark.parametrize("form", ["auto", "flat"])
def test_thin_output_conv_matches_the_grouped_kernel(n, cout, relu_cols, sizes, form, monkeypatch):
    """hn_conv3x3_thin_f16x3_levels (csrc/conv3x3_thin.hip), the FCOS head outputs (fcos_utils/fcos.py:247-264,299-320).
    Its tap kernel keeps the k and term order of the implicit GEMM: bit-identical.  Its P-form kernel (Cout <= 5, >= 64 k
    pixels -- or any size under the form "thin_form_flat", the second run) adds the same fp32 products in another order (per tap
    over all channels, then the nine taps): equal to fp32 rounding of a 2304-term sum, and as close to the fp64 convolution
    as the implicit GEMM is.  Inputs are channel slices of a 512-channel stack, as heads_grouped hands them over."""
    from hn_amd import ops
    from hn_amd.weights import ConvW
    ops.set_form("thin_form_flat", form == "flat")
    try:
        g = torch.Generator().manual_seed(1234 + n + cout)
        cin = 256
        w = (torch.randn(cout, 3, 3, cin, generator=g) * 0.05).cuda()
        b = torch.randn(cout, generator=g).cuda()
        cw = ConvW(w.cpu(), b.cpu(), 1, 1, 1).to("cuda")
        stacks = [ops.to_split(torch.randn(n, h, wd, 2 * cin, generator=g).cuda()) for h, wd in sizes]
        flat = ops.thin_uses_flat([s[:, :, :, :8] for s in stacks], cw)
        pixels = n * sum(h * wd for h, wd in sizes)
        assert flat == (cout <= 5 and max(wd for _, wd in sizes) <= 167 and (form == "flat" or pixels >= 65536))
        for half in (0, 1):
            xs = [s[:, :, :, 8 * half:8 * half + 8] for s in stacks]
            want = [ops.conv2d_nhwc(x, cw.w, cw.bias, pad=1, relu_cols=relu_cols, w16=cw.w16, splitk=False) for x in xs]
            if len(xs) > 1:   # (a single member would fall through to the split-K plan of conv2d_nhwc: another summation order)
                for a, e in zip(ops.conv2d_nhwc_grouped(xs, [cw] * len(xs), pad=1, relu_cols=relu_cols), want):
                    assert torch.equal(a, e)
            got = ops.conv3x3_thin_levels(xs, cw, relu_cols=relu_cols)
            for a, e, x in zip(got, want, xs):
                assert a.shape == e.shape
                if not flat:
                    assert torch.equal(a, e)
                    continue
                x64 = ops.from_split(x).double().permute(0, 3, 1, 2)
                ref = torch.nn.functional.conv2d(x64, w.double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
                ref = torch.cat([ref[..., :relu_cols].clamp_min(0), ref[..., relu_cols:]], dim=-1)
                scale = ref.abs().max().item()
                err_thin, err_gemm = (a.double() - ref).abs().max().item(), (e.double() - ref).abs().max().item()
                assert err_thin <= 4e-6 * scale and err_thin <= 2 * err_gemm + 1e-7 * scale, (err_thin, err_gemm, scale)
                assert (a - e).abs().max().item() <= 4e-6 * scale
        dense = [s[:, :, :, :8].contiguous() for s in stacks]
        for a, g2 in zip(ops.conv3x3_thin_levels(dense, cw, relu_cols=relu_cols),
                         ops.conv3x3_thin_levels([s[:, :, :, :8] for s in stacks], cw, relu_cols=relu_cols)):
            assert torch.equal(a, g2)     # dense and sliced inputs: the same kernel, the same bits
    finally:
        ops.set_form("thin_form_flat", False)


@pytest.mark.parametrize("n,sizes", [
    (1, [(100, 136), (50, 68), (25, 34)]),          # a single frame's pyramid: the tap kernel, ONE launch for the members
    (2, [(13, 17), (1, 2), (7, 300)]),              # odd maps, a level wider than the P form takes
    (8, [(100, 136), (50, 68), (25, 34)]),          # > 64 k pixels: the 5-channel members take the P form -> member by member
])
def test_thin_output_group_equals_its_members(n, sizes):
    """hn_conv3x3_thin_f16x3_levels_group (the FCOS head outputs -- cls + lr, ext, reg + ctr -- as one launch where they run the
    tap kernel): two and three members with their own filter banks, output widths, ReLU columns and inputs (channel slices of
    one 512-channel stack, as heads_grouped hands them over) == the same calls one after the other, bit for bit; also with the
    grouping switched off (form "thin_no_group")."""
    from hn_amd import ops
    from hn_amd.weights import ConvW
    g = torch.Generator().manual_seed(4321 + n)
    cin = 256
    banks = [ConvW(torch.randn(c, 3, 3, cin, generator=g) * 0.05, torch.randn(c, generator=g), 1, 1, 1).to("cuda") for c in (5, 8, 5)]
    stacks = [ops.to_split(torch.randn(n, h, wd, 2 * cin, generator=g).cuda()) for h, wd in sizes]
    ac, ar = [s[:, :, :, :8] for s in stacks], [s[:, :, :, 8:] for s in stacks]
    for members in ([(ac, banks[0], 0), (ar, banks[2], 4)], [(ac, banks[0], 0), (ac, banks[1], 3), (ar, banks[2], 4)],
                    [(ar, banks[1], 8)]):
        want = [ops.conv3x3_thin_levels(xs, cw, relu_cols=rc) for xs, cw, rc in members]
        for no_group in (False, True):
            ops.set_form("thin_no_group", no_group)
            try:
                got = ops.conv3x3_thin_levels_group(members)
            finally:
                ops.set_form("thin_no_group", False)
            assert len(got) == len(want)
            for gm, wm in zip(got, want):
                assert len(gm) == len(wm) and all(torch.equal(a, b) for a, b in zip(gm, wm))
    with pytest.raises(ValueError):
        ops.conv3x3_thin_levels_group([])
    with pytest.raises(ValueError):
        ops.conv3x3_thin_levels_group([(ac, banks[0], 0)] * 4)


def test_thin_output_conv_is_deterministic_at_full_size():
    """The P-form kernel hands P columns from 16 waves to the output threads through an LDS ring that it overwrites every
    step; a missing barrier would show as run-to-run differences.  Batch-32 pyramid (580 k pixels, 254 workgroup ranges
    sweeping ~10 steps each), 20 launches, all bit-identical; and the result equals the tap kernel's to fp32 rounding."""
    from hn_amd import ops
    from hn_amd.weights import ConvW
    g = torch.Generator().manual_seed(77)
    cw = ConvW(torch.randn(5, 3, 3, 256, generator=g) * 0.05, torch.randn(5, generator=g), 1, 1, 1).to("cuda")
    xs = [ops.to_split(torch.randn(32, h, w, 256, generator=g).cuda()) for h, w in ((100, 136), (50, 68), (25, 34), (13, 17), (7, 9))]
    assert ops.thin_uses_flat(xs, cw)
    first = ops.conv3x3_thin_levels(xs, cw, relu_cols=4)
    for _ in range(19):
        again = ops.conv3x3_thin_levels(xs, cw, relu_cols=4)
        assert all(torch.equal(a, b) for a, b in zip(first, again))
    ops.set_form("thin_form_tap", True)
    try:
        assert not ops.thin_uses_flat(xs, cw)
        tap = ops.conv3x3_thin_levels(xs, cw, relu_cols=4)
    finally:
        ops.set_form("thin_form_tap", False)
    for a, b in zip(first, tap):
        assert (a - b).abs().max().item() <= 4e-6 * b.abs().max().item()


@pytest.mark.parametrize("n,sizes", [
    (8, [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)]),      # the FCOS pyramid, 145 k pixels
    (300, [(13, 17), (7, 9)]),                                     # small maps only: every pixel range spans several images
    (33, [(25, 34), (50, 68)]),
])
def test_fused_groupnorm_apply_in_the_head_output_kernel_is_bit_identical(n, sizes):
    """hn_conv3x3_thin_affine_f16x3_levels: relu(x * scale + shift) and the fp16 hi / lo split happen on the P-form kernel's
    fragments (scale / shift rows of the images a workgroup touches in LDS).  Same arithmetic as hn_affine_split_f32_levels
    followed by hn_conv3x3_thin_f16x3_levels, expression for expression: torch.equal, for both channel halves of the
    512-channel tower output (fcos.py:232-239,247-264)."""
    from hn_amd import ops
    from hn_amd.weights import ConvW
    g = torch.Generator().manual_seed(4321 + n)
    cw = ConvW(torch.randn(5, 3, 3, 256, generator=g) * 0.05, torch.randn(5, generator=g), 1, 1, 1).to("cuda")
    ts = [(torch.randn(n, h, w, 512, generator=g) * 3).cuda() for h, w in sizes]
    aff = [((torch.rand(n, 512, generator=g) + 0.5).cuda(), torch.randn(n, 512, generator=g).cuda()) for _ in sizes]
    assert ops.thin_affine_applies(ts, cw)
    a = ops.to_split_levels(ts, aff, relu=True)
    for ch0, rc in ((0, 0), (256, 4)):
        want = ops.conv3x3_thin_levels([x[:, :, :, ch0 // 32:ch0 // 32 + 8] for x in a], cw, relu_cols=rc)
        got = ops.conv3x3_thin_affine_levels(ts, aff, ch0, cw, relu_cols=rc)
        for u, v in zip(got, want):
            assert u.shape == v.shape and torch.equal(u, v)
    # too small a problem (the tap kernel's territory) or maps too small for the table: the query says no
    assert not ops.thin_affine_applies([t[:1] for t in ts], cw)


def test_fused_groupnorm_apply_on_a_wide_frame_pyramid():
    """A 1280 x 720 frame resizes to 1344 x 768 (levels 96 x 168, 48 x 84, 24 x 42): the P-form ring for 168-wide rows leaves
    room for fewer scale / shift rows in LDS (the table holds what fits, three images at least); still bit-identical."""
    from hn_amd import ops
    from hn_amd.weights import ConvW
    g = torch.Generator().manual_seed(99)
    cw = ConvW(torch.randn(5, 3, 3, 256, generator=g) * 0.05, torch.randn(5, generator=g), 1, 1, 1).to("cuda")
    n, sizes = 6, [(96, 160), (48, 80), (24, 40)]
    ts = [(torch.randn(n, h, w, 512, generator=g) * 3).cuda() for h, w in sizes]
    aff = [((torch.rand(n, 512, generator=g) + 0.5).cuda(), torch.randn(n, 512, generator=g).cuda()) for _ in sizes]
    assert ops.thin_affine_applies(ts, cw)
    a = ops.to_split_levels(ts, aff, relu=True)
    want = ops.conv3x3_thin_levels([x[:, :, :, 8:] for x in a], cw, relu_cols=4)
    got = ops.conv3x3_thin_affine_levels(ts, aff, 256, cw, relu_cols=4)
    for u, v in zip(got, want):
        assert torch.equal(u, v)


def _multi_member(rng_seed, n, h, w, cin, cout, r, stride=1, dil=1, relu=True, residual=False, out_split=True):
    """(x S32, ConvW, opts) with random data; pad = the 'same' padding of the filter"""
    from hn_amd import ops
    from hn_amd.weights import ConvW
    g = torch.Generator().manual_seed(rng_seed)
    x = torch.randn((n, h, w, cin), generator=g).cuda()
    wt = (torch.randn((cout, r, r, cin), generator=g) * (2.0 / (r * r * cin)) ** 0.5)
    cw = ConvW(wt, torch.randn((cout,), generator=g) * 0.1, stride, dil * (r // 2), dil).to("cuda")
    oh, ow = ops.conv_out_size(h, w, r, r, stride, cw.pad, dil)
    res = ops.to_split(torch.randn((n, oh, ow, cout), generator=g).cuda()) if residual else None
    return ops.to_split(x), cw, dict(relu=relu, residual=res, out_split=out_split)


@pytest.mark.parametrize("case", ["bottleneck_ds", "basic_ds", "layer4_cls", "four", "mixed_tiles", "big_and_small"])
def test_multi_launch_is_bit_identical_to_separate_launches(case):
    """hn_conv2d_nhwc_f16x3_multi: independent convolutions of DIFFERENT shapes in one grid (the 1x1 downsample beside conv1 of
    a residual block, the classification head beside layer4) must return, member for member, the very bits of separate
    hn_conv2d_nhwc_f16x3_ws calls -- split-K members (with ONE reduction launch for all of them), residual / ReLU epilogues,
    fp32 and S32 outputs, strides and dilation included; members of different tile forms fall back to separate launches."""
    from hn_amd import ops
    members = {
        # A2J block 0 of layer2 at batch 1: conv1 1x1 256->128 beside the downsample 1x1 / stride 2 256->512
        "bottleneck_ds": [_multi_member(1, 1, 44, 44, 256, 128, 1), _multi_member(2, 1, 44, 44, 256, 512, 1, stride=2, relu=False)],
        # ResNet-34 layer3 block 0 at batch 1: conv1 3x3 / stride 2 128->256 beside the downsample 1x1 / stride 2
        "basic_ds": [_multi_member(3, 1, 100, 136, 128, 256, 3, stride=2), _multi_member(4, 1, 100, 136, 128, 256, 1, stride=2, relu=False)],
        # A2J layer4 block 1 conv2 (dilated) beside the classification head's first conv, batch 2; one member with a residual
        "layer4_cls": [_multi_member(5, 2, 11, 11, 512, 512, 3, dil=2), _multi_member(6, 2, 11, 11, 1024, 256, 3),
                       _multi_member(7, 2, 11, 11, 2048, 512, 1, residual=True)],
        "four": [_multi_member(8, 1, 22, 22, 128, 128, 3), _multi_member(9, 1, 22, 22, 512, 128, 1),
                 _multi_member(10, 1, 11, 11, 256, 336, 3, relu=False, out_split=False), _multi_member(11, 1, 22, 22, 128, 512, 1)],
        # a 128x128-tile member and a small one: different tile forms -> the library issues them separately
        "mixed_tiles": [_multi_member(12, 8, 100, 136, 128, 256, 3), _multi_member(13, 1, 11, 11, 256, 256, 3)],
        "big_and_small": [_multi_member(14, 16, 22, 22, 128, 128, 3), _multi_member(15, 16, 22, 22, 512, 128, 1)],
    }[case]
    ref = [ops.conv2d_nhwc(x, cw.w, cw.bias, stride=cw.stride, pad=cw.pad, dil=cw.dil, w16=cw.w16, relu=o["relu"],
                           residual=o["residual"], out_split=o["out_split"]) for x, cw, o in members]
    fused = []
    got = ops.conv2d_nhwc_multi(members, fused=fused)
    assert fused[0] == (case != "mixed_tiles"), (case, fused)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), case
    # against fp64 as well (the separate launches are themselves checked elsewhere; this pins the multi path on its own)
    x, cw, o = members[0]
    xf = ops.from_split(x).double().permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(xf, cw.w.double().permute(0, 3, 1, 2), cw.bias.double(), stride=cw.stride, padding=cw.pad,
                                   dilation=cw.dil)
    if o["residual"] is not None:
        y = y + ops.from_split(o["residual"]).double().permute(0, 3, 1, 2)
    if o["relu"]:
        y = y.relu()
    out = got[0] if not o["out_split"] else ops.from_split(got[0])
    assert (out.double().permute(0, 3, 1, 2) - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())


def test_halo_shape_with_unaligned_operands_takes_the_implicit_gemm():
    """A call whose SHAPE routes to the halo-patch kernel (64 -> 64 3x3, >= 512 tiles) but whose bias is not 16-byte aligned must
    not fail: it falls through to the implicit-GEMM kernels, with the same bits as the aligned call."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    n, h, w = 2, 200, 272
    x = ops.to_split(_rand((n, h, w, 64), 91).cuda())
    wt = _rand((64, 3, 3, 64), 92, (2.0 / 576) ** 0.5)
    w16 = split_f16x3(wt).cuda()
    pad = torch.zeros((65,), device="cuda")
    pad[1:] = _rand((64,), 93, 0.1).cuda()
    bias_unaligned = pad[1:]                       # 4 bytes past a 16-byte boundary
    assert bias_unaligned.data_ptr() % 16 == 4
    bias_aligned = bias_unaligned.clone()
    y0 = ops.conv2d_nhwc(x, wt.cuda(), bias_aligned, pad=1, relu=True, w16=w16, out_split=True)
    y1 = ops.conv2d_nhwc(x, wt.cuda(), bias_unaligned, pad=1, relu=True, w16=w16, out_split=True)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, residual kind, out_split, relu)
    (5, 100, 136, 128, 256, "up_s32", True, False),     # the FPN P3 lateral: top-down add read at (h/2, w/2) (res_mode 2), S32 in / out
    (34, 44, 44, 64, 256, "same_s32", True, True),      # A2J layer1 expansion + identity + ReLU; 65 824 pixels: a ragged last tile
    (140, 22, 22, 128, 512, None, False, False),        # two 256-channel column groups, fp32 output, no residual
    (36, 44, 44, 64, 256, "same_f32", True, True),      # fp32 residual
])
def test_streaming_1x1_kernel_is_bit_identical_to_the_implicit_gemm(case):
    """conv1x1_stream_kernel (1x1 / stride 1, Cin 64 or 128, Cout % 256 == 0, >= 65 536 pixels) against the
    implicit-GEMM kernel it replaces for that shape (form "conv_no_stream"): same k order, same term order, same epilogue
    arithmetic -- same bits; and against an fp64 convolution."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    from oracle import ops_ref
    n, h, w, cin, cout, res, osplit, relu = case
    x = _rand((n, h, w, cin), 91)
    wt = _rand((cout, 1, 1, cin), 92, scale=(2.0 / cin) ** 0.5)
    b = _rand((cout,), 93, 0.1)
    r = None
    if res == "up_s32":
        r = _rand((n, h // 2, w // 2, cout), 94)
    elif res is not None:
        r = _rand((n, h, w, cout), 94)
    xs, w16 = ops.to_split(x.cuda()), split_f16x3(wt).cuda()
    rdev = None if r is None else (r.cuda() if res == "same_f32" else ops.to_split(r.cuda()))
    d = ops.make_conv_desc(n, h, w, cin, cout, 1, 1, 1, 0, 1, cout if relu else 0, 0 if res is None else (2 if res == "up_s32" else 1))
    d.out_split = 1 if osplit else 0
    assert ops._lib.load().hn_conv2d_f16x3_uses_stream(d) == 1
    kw = dict(pad=0, relu=relu, w16=w16, out_split=osplit, residual=rdev, res_upsample=res == "up_s32")
    block = torch.zeros((4,), device="cuda", dtype=torch.int32)
    with ops.range_scope(block):
        y_stream = ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), **kw)
        assert ops.range_check_collect(block).cpu().tolist() == [0, 0, 0, 0]
    ops.set_form("conv_no_stream", True)
    try:
        assert ops._lib.load().hn_conv2d_f16x3_uses_stream(d) == 0
        y_gemm = ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), **kw)
    finally:
        ops.set_form("conv_no_stream", False)
    assert y_stream.shape == y_gemm.shape and torch.equal(y_stream, y_gemm)
    ref = ops_ref.conv2d_nhwc(x.double(), wt.double(), b.double(), 1, 0, 1, relu_cols=0)
    if res == "up_s32":
        ref = ref + r.double().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    elif res is not None:
        ref = ref + r.double()
    ref = (torch.relu(ref) if relu else ref).float()
    got = (ops.from_split(y_stream) if osplit else y_stream).cpu()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # the range contract rides along: an overflowing S32 output is flagged
    if osplit:
        big = wt * 3.0e4
        with ops.range_scope(block):
            ops.conv2d_nhwc(xs, big.cuda(), b.cuda(), **dict(kw, w16=split_f16x3(big).cuda()))
            assert ops.range_check_collect(block).cpu().tolist() == [1, 0, 0, 0]


def test_streaming_1x1_kernel_leaves_small_and_odd_shapes_to_the_implicit_gemm():
    from hn_amd import ops
    lib = ops._lib.load()

    def uses(n, h, w, cin, cout, r=1, stride=1, relu_cols=0):
        d = ops.make_conv_desc(n, h, w, cin, cout, r, r, stride, r // 2, 1, relu_cols, 0)
        return lib.hn_conv2d_f16x3_uses_stream(d)
    assert uses(32, 100, 136, 128, 256) == 1 and uses(64, 44, 44, 64, 256) == 1
    assert uses(1, 100, 136, 128, 256) == 0          # batch 1: 13 600 pixels do not fill the chip with 128-pixel tiles
    assert uses(32, 100, 136, 256, 256) == 0         # k too long for register-resident filters
    assert uses(32, 100, 136, 128, 128) == 0         # not a whole 256-channel column group
    assert uses(32, 100, 136, 128, 256, r=3) == 0 and uses(32, 100, 136, 128, 256, stride=2) == 0
    assert uses(32, 100, 136, 128, 256, relu_cols=100) == 0


@pytest.mark.parametrize("n", [1, 2])
def test_mixed_tile_grouped_launch_is_bit_identical_to_one_tile_shape(n):
    """conv_igemm_f16x3_mixed_kernel (round 5): a grouped launch over the FPN levels of both towers whose last round of 128 x 128
    tiles would be nearly empty (batch 1: 564 tiles on 512 slots; batch 2: 1128 = 2.2 rounds) gives the stride-16 / stride-32
    members 64 x 128 per-tap tiles while the stride-8 members keep the 128 x 128 row-shared form.  Same k order: outputs and the
    GroupNorm partial sums of the stacked slab are bit-identical to the plain grouped launch (form "conv_no_mixed"), as S32
    with ReLU and as fp32 with the GroupNorm sums (what the towers use)."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(97)
    dims = [(100, 136), (50, 68), (25, 34)]
    L = len(dims)
    xs = [ops.to_split(torch.randn((n, h, w, 512), generator=g).cuda()) for h, w in dims]
    cws = []
    for _ in range(2):
        wt = torch.randn((256, 3, 3, 256), generator=g) * (2.0 / 2304) ** 0.5
        cws.append(NS(w=wt.cuda(), bias=torch.randn((256,), generator=g).cuda(), w16=split_f16x3(wt).cuda()))
    members = [x[:, :, :, :8] for x in xs] + [x[:, :, :, 8:] for x in xs]
    weights = [cws[0]] * L + [cws[1]] * L

    def run():
        hw = [h * w for h, w in dims]
        parts = [torch.full((ops.gn_rows32_scratch_floats(n * hw[l], 512),), float("nan"), device="cuda") for l in range(L)]
        t = [torch.empty((n, h, w, 512), device="cuda") for h, w in dims]
        ops.conv2d_nhwc_grouped(members, weights, pad=1, outs=t + t, out_channel_offsets=[0] * L + [256] * L,
                                gn=[(parts[l], 0) for l in range(L)] + [(parts[l], 32) for l in range(L)], gn_units=64)
        s = ops.conv2d_nhwc_grouped(members, weights, pad=1, relu=True, out_split=True)
        return t, parts, s
    t_mixed, p_mixed, s_mixed = run()
    ops.set_form("conv_no_mixed", True)
    try:
        t_plain, p_plain, s_plain = run()
    finally:
        ops.set_form("conv_no_mixed", False)
    for a, b in zip(t_mixed + p_mixed + s_mixed, t_plain + p_plain + s_plain):
        assert not torch.isnan(a).any() and torch.equal(a, b)
    # and the plain form on one member alone agrees (the grouped launch is not its own only reference)
    ref = ops.conv2d_nhwc(members[1], cws[0].w, cws[0].bias, pad=1, relu=True, w16=cws[0].w16, out_split=True, splitk=False)
    assert torch.equal(s_mixed[1], ref)


DEEPK_CASES = [
    # n, h, w, cin, cout, r, stride, pad, dil
    (1, 50, 68, 256, 256, 3, 1, 1, 1),      # ResNet-34 layer3 at batch 1: 72 k tiles
    (1, 100, 136, 128, 128, 3, 1, 1, 1),    # layer2
    (1, 25, 34, 512, 512, 3, 1, 1, 1),      # layer4 (split-K in the auto plan)
    (1, 11, 11, 512, 512, 3, 1, 2, 2),      # A2J layer4, dilation 2
    (2, 11, 11, 1024, 256, 1, 1, 0, 1),     # 1x1, 32 k tiles
    (1, 44, 44, 128, 128, 3, 2, 1, 1),      # stride 2
    (3, 13, 17, 32, 48, 3, 1, 1, 1),        # 9 k tiles: a ragged last stage; ragged rows and columns
    (1, 9, 9, 32, 64, 1, 1, 0, 1),          # ONE k tile: fewer tiles than a stage holds
    (2, 22, 22, 96, 80, 3, 1, 1, 1),        # 27 k tiles
]


@pytest.mark.parametrize("case", DEEPK_CASES)
@pytest.mark.parametrize("tiles", [(3, 12)])
def test_deep_k_forms_are_bit_identical_to_their_parent_tiles(case, tiles):
    """The deep-k loop (two k tiles per ring stage, one barrier per stage, compiler-scheduled; HN_TILE_64x64_K2) keeps the pinned
    loop's DMA geometry, k order and term order: S32 / fp32 outputs, residual + ReLU, with and without split-K, equal the parent
    tile's bit for bit -- including k loops with an odd tile count and with fewer tiles than one stage."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    n, h, w, cin, cout, r, stride, pad, dil = case
    parent, deep = tiles
    x = _rand((n, h, w, cin), 101)
    wt = _rand((cout, r, r, cin), 102, scale=(2.0 / (cin * r * r)) ** 0.5)
    b = _rand((cout,), 103, 0.1)
    oh, ow = ops.conv_out_size(h, w, r, r, stride, pad, dil)
    res = _rand((n, oh, ow, cout), 104)
    xs, w16 = ops.to_split(x.cuda()), split_f16x3(wt).cuda()
    outs = {}
    for tile in (parent, deep):
        kw = dict(stride=stride, pad=pad, dil=dil, relu=True, tile=tile, w16=w16)
        got = [ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), splitk=False, **kw)]
        if cout % 32 == 0:
            got.append(ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), residual=ops.to_split(res.cuda()), out_split=True, splitk=False, **kw))
        if cout % 8 == 0 and r * r * cin // 32 >= 4:
            got.append(ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), force_splits=2, **kw))
            got.append(ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), force_splits=3, **kw))
        outs[tile] = got
    for a, c in zip(outs[parent], outs[deep]):
        assert a.shape == c.shape and torch.equal(a, c), (case, tiles)


def test_deep_k_tile_pick_and_multi_launch():
    """The auto heuristic picks the deep-k form of the 64x64 tile for grids of at most one workgroup per CU with >= 8 k tiles
    (ResNet-34 layer3 at batch 1), never at batch 32 and never with "conv_no_deepk"; a heterogeneous launch whose members take the
    64x64 tile -- deep-k ones and one whose k loop is too short for it on its own -- is one launch, bit-identical to separate ones."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    lib = ops._lib.load()

    def pick(n, h, w, cin, cout, r, stride=1):
        return lib.hn_conv2d_f16x3_pick_tile(ops.make_conv_desc(n, h, w, cin, cout, r, r, stride, r // 2, 1, cout, 0))
    assert pick(1, 50, 68, 256, 256, 3) == 12 and pick(1, 50, 68, 256, 256, 1) == 12
    assert pick(1, 100, 136, 128, 128, 3) == 6 and pick(1, 11, 11, 512, 512, 3) == 7       # the other small tiles keep the pinned loop
    assert pick(32, 50, 68, 256, 256, 3) == 1 and pick(1, 50, 68, 128, 256, 1) == 3       # batch 32; 4 k tiles only
    ops.set_form("conv_no_deepk", True)
    try:
        assert pick(1, 50, 68, 256, 256, 3) == 3
    finally:
        ops.set_form("conv_no_deepk", False)
    g = torch.Generator().manual_seed(105)
    items = []
    for cin, cout, r in ((256, 256, 3), (256, 256, 1), (128, 256, 1)):
        x = ops.to_split(torch.randn((1, 50, 68, cin), generator=g).cuda())
        wt = torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5
        cw = NS(w=wt.cuda(), bias=torch.randn((cout,), generator=g).cuda(), w16=split_f16x3(wt).cuda(), stride=1, pad=r // 2, dil=1)
        items.append((x, cw, dict(relu=r == 3, residual=None, out_split=True)))
    alone = [ops.conv2d_nhwc(xi, cw.w, cw.bias, stride=cw.stride, pad=cw.pad, dil=cw.dil, w16=cw.w16, **o) for xi, cw, o in items]
    flag = []
    fused = ops.conv2d_nhwc_multi(items, fused=flag)
    assert flag[0]
    for a, c in zip(fused, alone):
        assert torch.equal(a, c)


@pytest.mark.parametrize("case,tile,splits", [
    ((1, 25, 34, 512, 512, 3, 1, 1, 1), 7, 3), ((1, 25, 34, 512, 512, 3, 1, 1, 1), 7, 16), ((1, 11, 11, 512, 512, 3, 1, 1, 1), 7, 5),
    ((1, 11, 11, 2048, 512, 3, 1, 1, 1), 7, 8), ((1, 11, 11, 1024, 256, 1, 1, 0, 1), 7, 2), ((1, 50, 68, 256, 256, 3, 1, 1, 1), 3, 2),
    ((1, 50, 68, 256, 256, 3, 1, 1, 1), 12, 3), ((1, 100, 136, 128, 128, 3, 1, 1, 1), 6, 2), ((2, 23, 19, 160, 72, 3, 2, 1, 1), 3, 3),
    ((1, 11, 11, 512, 512, 3, 1, 2, 2), 6, 4), ((32, 11, 11, 512, 512, 3, 1, 1, 1), 6, 12)])
def test_split_k_reduction_in_the_last_workgroup_is_bit_identical(case, tile, splits):
    """Round 5: a split-K launch of the 32- / 64-row tiles reduces in the workgroup that finishes a tile LAST (a ticket per tile and
    wave, zero at rest) instead of in a second launch.  The planes are added in z order whoever arrives last, so fp32 and S32 outputs,
    with residual and ReLU, equal the separate reduction ("conv_no_fused_reduce") bit for bit -- on every one of many repeats
    (which workgroup reduces differs from run to run), and the tickets are zero again for the next launch."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    n, h, w, cin, cout, r, stride, pad, dil = case
    x = _rand((n, h, w, cin), 111)
    wt = _rand((cout, r, r, cin), 112, scale=(2.0 / (cin * r * r)) ** 0.5)
    b = _rand((cout,), 113, 0.1)
    oh, ow = ops.conv_out_size(h, w, r, r, stride, pad, dil)
    res = ops.to_split(_rand((n, oh, ow, cout), 114).cuda()) if cout % 32 == 0 else None
    xs, w16 = ops.to_split(x.cuda()), split_f16x3(wt).cuda()
    kw = dict(stride=stride, pad=pad, dil=dil, relu=True, tile=tile, w16=w16, force_splits=splits)

    def run():
        out = [ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), **kw)]
        if cout % 32 == 0:
            out.append(ops.conv2d_nhwc(xs, wt.cuda(), b.cuda(), residual=res, out_split=True, **kw))
        return out
    ops.set_form("conv_no_fused_reduce", True)
    try:
        ref = run()
    finally:
        ops.set_form("conv_no_fused_reduce", False)
    for rep in range(60 if splits == 16 else 12):   # (the 16-way split of 112 tiles: 1792 workgroups racing for 448 wave tickets)
        for a, c in zip(ref, run()):
            assert torch.equal(a, c), (case, tile, splits, rep)


def test_split_k_in_kernel_reduction_of_a_heterogeneous_launch():
    """Members of one multi launch that split keep their own ticket ranges: two 11 x 11 layers of the 32x64 tile (both split by
    the plan) beside each other equal their separate-reduction results bit for bit, repeatedly."""
    from types import SimpleNamespace as NS
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    g = torch.Generator().manual_seed(115)
    items = []
    for cin, cout, r in ((512, 512, 3), (1024, 512, 3), (2048, 256, 1)):
        x = ops.to_split(torch.randn((1, 11, 11, cin), generator=g).cuda())
        wt = torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5
        cw = NS(w=wt.cuda(), bias=torch.randn((cout,), generator=g).cuda(), w16=split_f16x3(wt).cuda(), stride=1, pad=r // 2, dil=1)
        items.append((x, cw, dict(relu=True, residual=None, out_split=True)))
    ops.set_form("conv_no_fused_reduce", True)
    try:
        flag = []
        ref = ops.conv2d_nhwc_multi(items, fused=flag)
    finally:
        ops.set_form("conv_no_fused_reduce", False)
    assert flag[0]
    for rep in range(12):
        flag = []
        got = ops.conv2d_nhwc_multi(items, fused=flag)
        assert flag[0]
        for a, c in zip(ref, got):
            assert torch.equal(a, c), rep


def test_split_k_tickets_with_more_workspaces_than_slots():
    """The library keeps one ticket slot per workspace ADDRESS (128 of them, never re-assigned); a process that splits with more
    addresses than that gets the separate reduction launch for the later ones -- same bits (tests/ticket_registry_worker.py, in
    its own process: it uses the registry up)."""
    import subprocess
    import sys
    from pathlib import Path
    r = subprocess.run([sys.executable, str(Path(__file__).with_name("ticket_registry_worker.py"))], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "mismatching launches: 0" in r.stdout, r.stdout + r.stderr

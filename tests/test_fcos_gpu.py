"""FCOS stages on HIP vs the oracle (staged, as SURVEY 7 'detection-stage chaos' prescribes:
each stage is fed the ORACLE's inputs so one borderline threshold flip cannot cascade).

Bars: integer / index outputs bit-exact; fp32 tensors within the tolerance written next to
each assert.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle_run(fcos_sd):
    from hn_amd import synth
    from oracle import fcos_ref
    rgb = synth.make_rgb(2, seed=1000)
    dets, inter = fcos_ref.fcos_forward([rgb[0], rgb[1]], fcos_sd, 3, return_intermediates=True)
    return rgb, dets, inter


@pytest.fixture(scope="module")
def engine(fcos_sd):
    from hn_amd.fcos_engine import FCOSEngine
    return FCOSEngine(fcos_sd, 3, device="cuda")


def test_preprocess_matches_transform(oracle_run, engine):
    from hn_amd import ops
    from hn_amd.fcos_engine import IMAGE_MEAN, IMAGE_STD
    rgb, _, inter = oracle_run
    oh, ow, ph, pw = engine.geometry(480, 640)
    assert (oh, ow, ph, pw) == (800, 1066, 800, 1088) == (*inter["image_sizes"][0], *inter["x"].shape[-2:])
    y = ops.fcos_preprocess(rgb.cuda(), oh, ow, ph, pw, IMAGE_MEAN, IMAGE_STD).cpu()
    ref = inter["x"].permute(0, 2, 3, 1)
    assert float(y[..., 3].abs().max()) == 0.0
    # fp32 bilinear with the source-index arithmetic AND the fused multiply-add order of ATen's CPU kernel: bit-identical
    assert torch.equal(y[..., :3], ref)
    x16 = ops.fcos_preprocess_split(rgb.cuda(), oh, ow, ph, pw, IMAGE_MEAN, IMAGE_STD)      # the tiled split form
    hi = x16[0, :, 3:-3, 3:-3, :3].float().cpu()
    assert torch.equal(hi, ref.half().float())


def test_split_stem_matches_f32_path(oracle_run, engine, fcos_sd):
    """The f16x3 stem (split image with a zero border, one filter row per k tile) vs the same arithmetic in
    fp64 on the host: preprocess planes recombine to the fp32 canvas to 2^-21 relative, conv1+bn1+ReLU to 2e-5."""
    import torch.nn.functional as F
    from hn_amd import ops
    from hn_amd.fcos_engine import IMAGE_MEAN, IMAGE_STD
    from hn_amd.weights import bn_scale_shift
    rgb, _, inter = oracle_run
    x32 = ops.fcos_preprocess(rgb.cuda(), 800, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
    x16 = ops.fcos_preprocess_split(rgb.cuda(), 800, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
    assert tuple(x16.shape) == (2, rgb.shape[0], 806, 1094, 4)
    rec = x16[0].float() + x16[1].float()
    inner = rec[:, 3:-3, 3:-3]
    assert (inner - x32).abs().max().item() <= 2.0 ** -21 * x32.abs().max().item() + 3e-8
    border = rec.clone()
    border[:, 3:-3, 3:-3] = 0
    assert float(border.abs().max()) == 0.0
    y = ops.from_split(ops.conv_stem_split(x16, engine.stem16.w16, engine.stem16.bias, 64))
    scale, shift = bn_scale_shift(fcos_sd, "backbone.body.bn1")
    ref = F.conv2d(x32.cpu().double().permute(0, 3, 1, 2)[:, :3], fcos_sd["backbone.body.conv1.weight"].double(),
                   stride=2, padding=3)
    ref = torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert tuple(y.shape) == tuple(ref.shape) == (rgb.shape[0], 400, 544, 64)
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_backbone_and_heads_match_oracle(oracle_run, engine):
    rgb, _, inter = oracle_run
    from hn_amd import ops
    from hn_amd.fcos_engine import IMAGE_MEAN, IMAGE_STD
    x = ops.fcos_preprocess_split(rgb.cuda(), 800, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
    feats = engine.backbone(x)
    for f, rf in zip(feats, inter["features"]):
        f32 = ops.from_split(f) if ops.is_split(f) else f
        d = (f32.cpu() - rf.permute(0, 2, 3, 1)).abs().max().item()
        assert d <= 2e-4 * max(1.0, rf.abs().max().item()), d
    ho = inter["head"]
    start = 0
    for f in feats:
        cls_lr, reg_ctr, _ = engine.head_level(f)
        n, h, w, _ = cls_lr.shape
        assert cls_lr.dtype == torch.float32 and reg_ctr.dtype == torch.float32
        sl = slice(start, start + h * w)
        got_cls = cls_lr.cpu().reshape(n, h * w, 5)
        got_reg = reg_ctr.cpu().reshape(n, h * w, 5)
        # logits: tolerance 5e-4 absolute on O(1..10) values (4 GN layers deep)
        assert (got_cls[..., :3] - ho["cls_logits"][:, sl]).abs().max().item() <= 5e-4
        assert (got_cls[..., 3:] - ho["hand_lr"][:, sl]).abs().max().item() <= 5e-4
        assert (got_reg[..., :4] - ho["bbox_regression"][:, sl]).abs().max().item() <= 5e-4
        assert (got_reg[..., 4:] - ho["bbox_ctrness"][:, sl]).abs().max().item() <= 5e-4
        start += h * w
    assert start == 17850


def _oracle_heads_as_levels(inter, n):
    """Oracle head tensors [N,17850,k] -> per-level NHWC tensors the kernels consume."""
    ho = inter["head"]
    cls_lr = torch.cat([ho["cls_logits"], ho["hand_lr"]], -1)
    reg_ctr = torch.cat([ho["bbox_regression"], ho["bbox_ctrness"]], -1)
    levels, start = [], 0
    for f in inter["features"]:
        h, w = f.shape[-2:]
        levels.append((cls_lr[:, start:start + h * w].reshape(n, h, w, 5).contiguous().cuda(),
                       reg_ctr[:, start:start + h * w].reshape(n, h, w, 5).contiguous().cuda()))
        start += h * w
    return [l[0] for l in levels], [l[1] for l in levels]


def test_candidates_match_oracle_given_oracle_heads(oracle_run):
    from hn_amd import ops
    _, _, inter = oracle_run
    cls_lr, reg_ctr = _oracle_heads_as_levels(inter, 2)
    cand = ops.fcos_candidates(cls_lr, reg_ctr, [8, 16, 32], 3, 0.7)
    ho = inter["head"]
    all_scores = torch.sqrt(torch.sigmoid(ho["cls_logits"]) * torch.sigmoid(ho["bbox_ctrness"])).max(-1)[0]
    for i, ref in enumerate(inter["candidates"]):
        k = int(cand.count[i])
        borderline = int(((all_scores[i] - 0.7).abs() < 1e-6).sum())
        if borderline == 0:
            assert k == len(ref["scores"])
            assert torch.equal(cand.labels[i, :k].cpu().long(), ref["labels"])          # bit-exact ints
            assert torch.equal(cand.sides[i, :k].cpu().long(), ref["sides"])
            assert torch.equal(cand.level[i, :k].cpu().float(), ref["feature_idx"])
            assert torch.equal(cand.boxes[i, :k].cpu(), ref["boxes"])                     # decode is exact fp32
            assert (cand.scores[i, :k].cpu() - ref["scores"]).abs().max().item() <= 2e-7  # expf/sqrtf ulp
        else:
            assert abs(k - len(ref["scores"])) <= borderline


def _random_boxes(k, seed, spread=400.0, size=60.0):
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand((k, 2), generator=g) * spread
    wh = 5.0 + torch.rand((k, 2), generator=g) * size
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.rand((k,), generator=g) * 0.3 + 0.7
    labels = torch.randint(0, 3, (k,), generator=g)
    return boxes, scores, labels


@pytest.mark.parametrize("k", [0, 1, 7, 63, 64, 65, 128, 129, 257, 511, 512, 513, 1000, 1001, 1024, 1025, 2048, 2049, 2500, 5000])
def test_batched_nms_bit_exact(k):
    """Both torchvision paths (coordinate trick for K <= 1000, per-class above) and the
    LDS / global-memory sort paths (K <= 2048 / above)."""
    from hn_amd import ops
    from oracle import fcos_ref
    boxes, scores, labels = _random_boxes(max(k, 1), 100 + k)
    boxes, scores, labels = boxes[:k], scores[:k], labels[:k]
    ref_keep = fcos_ref.batched_nms(boxes, scores, labels, 0.3)
    cap = max(k, 8)
    cand = ops.alloc_candidates(1, cap, "cuda")
    cand.boxes[0, :k] = boxes.cuda()
    cand.scores[0, :k] = scores.cuda()
    cand.labels[0, :k] = labels.int().cuda()
    cand.count[0] = k
    det = ops.fcos_nms(cand, 0.3, 0.6, 0.6003752)
    n = int(det.count[0])
    assert n == len(ref_keep)
    assert torch.equal(det.keep[0, :n].cpu().long(), ref_keep)  # survivor indices, bit-exact, score order
    rw, rh = torch.tensor(0.6003752, dtype=torch.float32), torch.tensor(0.6, dtype=torch.float32)
    rb = boxes[ref_keep] * torch.stack([rw, rh, rw, rh])  # resize_boxes: one fp32 multiply per coord
    assert torch.equal(det.boxes[0, :n].cpu(), rb)
    assert torch.equal(det.scores[0, :n].cpu(), scores[ref_keep])
    assert torch.equal(det.labels[0, :n].cpu().long(), labels[ref_keep])


def _nms_case(name):
    """Edge fixtures for the greedy pass (inputs only; expectations come from oracle/nms_ref.c)."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    if name == "zero_area":
        # degenerate boxes: area 0 -> IoU of two of them is 0/0 = NaN, which is NOT > 0.3: nothing suppressed by
        # them, and a zero-area box inside a real one has IoU 0
        b, s, l = _random_boxes(200, 7)
        b[::5, 2] = b[::5, 0]            # zero width
        b[3::7, 3] = b[3::7, 1]          # zero height
        b[10] = b[15] = torch.tensor([50.0, 50.0, 50.0, 50.0])   # identical points
        return b, s, l
    if name == "identical_across_tile":
        # 70 copies of ONE box with ONE score and one class: they straddle the 64-candidate tile boundary of the
        # greedy pass; exactly the lowest index survives.  Then 70 more with distinct classes (all survive).
        b = torch.tensor([[10.0, 20.0, 90.0, 120.0]]).repeat(140, 1)
        s = torch.full((140,), 0.875)
        l = torch.cat([torch.zeros(70, dtype=torch.long), torch.arange(70) % 3 + 3])
        return b, s, l
    if name in ("ties_2049", "ties_4097"):
        k = int(name.split("_")[1])
        b, s, l = _random_boxes(k, 11 + k, spread=900.0)
        s = (torch.randint(0, 40, (k,), generator=g).float() / 64.0 + 0.3)   # ~40 distinct scores: long tie runs
        return b, s, l
    if name == "dense_chains_500":
        # 500 heavily overlapping boxes with distinct scores (the bitmask form's largest size class): long suppression chains --
        # A drops B, B would have dropped C but is gone, C must survive
        return _random_boxes(500, 77, spread=160.0, size=80.0)
    if name == "ties_500":
        b, s, l = _random_boxes(500, 78, spread=300.0, size=60.0)
        s = (torch.randint(0, 12, (500,), generator=g).float() / 16.0 + 0.2)   # 12 distinct scores: tie runs across the 64-bit words
        return b, s, l
    if name == "one_class_1001":
        b, s, l = _random_boxes(1001, 33)
        return b, s, torch.full((1001,), 2, dtype=torch.long)
    if name == "tie_run_over_tiles":
        # 200 overlapping boxes sharing one score: order within the run must be ascending index across tiles
        b, s, l = _random_boxes(200, 5, spread=60.0, size=80.0)
        return b, torch.full((200,), 0.75), torch.zeros(200, dtype=torch.long)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["zero_area", "identical_across_tile", "ties_2049", "ties_4097", "one_class_1001",
                                  "tie_run_over_tiles", "dense_chains_500", "ties_500"])
def test_batched_nms_edge_fixtures(name):
    """Survivor indices bit-exact vs oracle/nms_ref.c on the inputs where a greedy NMS can go wrong: NaN IoUs,
    equal (score, box) runs across the 64-candidate tiles, K just past the LDS sort capacity (2048) and past 4096
    with score ties (global-memory sort path + per-class torchvision path), and a single-class K = 1001."""
    from hn_amd import ops
    from oracle import fcos_ref
    boxes, scores, labels = _nms_case(name)
    k = boxes.shape[0]
    ref_keep = fcos_ref.batched_nms(boxes, scores, labels, 0.3)
    cand = ops.alloc_candidates(1, k, "cuda")
    cand.boxes[0] = boxes.cuda()
    cand.scores[0] = scores.cuda()
    cand.labels[0] = labels.int().cuda()
    cand.count[0] = k
    det = ops.fcos_nms(cand, 0.3, 1.0, 1.0)
    n = int(det.count[0])
    assert n == len(ref_keep), (n, len(ref_keep))
    assert torch.equal(det.keep[0, :n].cpu().long(), ref_keep)
    assert torch.equal(det.boxes[0, :n].cpu(), boxes[ref_keep])
    if name == "identical_across_tile":
        assert ref_keep[0] == 0 and n == 1 + 3   # one survivor of the 70 clones + one per distinct class


def test_batched_nms_at_full_capacity():
    """Maximum size: EVERY anchor point of a 480x640 frame is a candidate (K = capacity = 17850, the global-memory sort
    path with a 32768-key bitonic network and the per-class torchvision path), two images in one launch with
    different counts; survivor indices bit-exact vs oracle/nms_ref.c."""
    from hn_amd import ops
    from oracle import fcos_ref
    cap = 17850
    boxes, scores, labels = _random_boxes(cap, 77, spread=1000.0, size=90.0)
    counts = [cap, cap - 4097]
    cand = ops.alloc_candidates(2, cap, "cuda")
    for i, k in enumerate(counts):
        cand.boxes[i, :k] = boxes[:k].cuda()
        cand.scores[i, :k] = scores[:k].cuda()
        cand.labels[i, :k] = labels[:k].int().cuda()
        cand.count[i] = k
    det = ops.fcos_nms(cand, 0.3, 1.0, 1.0)
    for i, k in enumerate(counts):
        ref_keep = fcos_ref.batched_nms(boxes[:k], scores[:k], labels[:k], 0.3)
        n = int(det.count[i])
        assert n == len(ref_keep) and n > 500
        assert torch.equal(det.keep[i, :n].cpu().long(), ref_keep)


def test_plain_nms_zero_area_and_ties():
    from hn_amd import ops
    from oracle import fcos_ref
    for name in ("zero_area", "tie_run_over_tiles"):
        boxes, scores, _ = _nms_case(name)
        ref = fcos_ref.nms(boxes, scores, 0.3)
        got = ops.nms(boxes.cuda(), scores.cuda(), 0.3).cpu()
        assert torch.equal(got, ref), name


def test_nms_ties_and_threshold_boundary():
    from hn_amd import ops
    from oracle import fcos_ref
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0],
                          [0, 0, 10, 3.0]])  # last: IoU with box 0 is exactly 0.3 -> suppressed
    scores = torch.tensor([0.9, 0.8, 0.7, 0.9, 0.85])
    ref = fcos_ref.nms(boxes, scores, 0.3)
    got = ops.nms(boxes.cuda(), scores.cuda(), 0.3).cpu()
    assert got.tolist() == ref.tolist() == [0, 2]
    assert ops.nms(boxes[:0].cuda(), scores[:0].cuda(), 0.3).numel() == 0


def test_detect_end_to_end_agreement(oracle_run, engine):
    """Whole detector on HIP vs the oracle on the fixed seeds, EXACT (VERDICT r03 weak #1c; no matching rate):
    the candidate set (anchor-point indices) is identical; the reference NMS fed the oracle's boxes in the HIP engine's
    score order returns exactly the HIP survivor list; labels identical; and the survivor list equals the oracle's own
    unless two scores closer than twice the measured HIP-vs-oracle score difference swapped places (asserted on the gap)."""
    from oracle import fcos_ref
    rgb, dets, inter = oracle_run
    det, cand = engine.detect(rgb.cuda())
    for i, ref in enumerate(dets):
        oc = inter["candidates"][i]
        kc = int(cand.count[i])
        assert torch.equal(cand.point[i, :kc].cpu().long(), oc["index"])
        hs = cand.scores[i, :kc].cpu()
        delta = float((hs - oc["scores"]).abs().max())
        assert delta <= 3e-5
        k = int(det.count[i])
        hk = det.keep[i, :k].cpu().long()
        assert torch.equal(fcos_ref.batched_nms(oc["boxes"], hs, oc["labels"], 0.3), hk)
        assert torch.equal(det.labels[i, :k].cpu().long(), oc["labels"][hk])
        if not torch.equal(hk, ref["keep"]):
            os_ = oc["scores"]
            pos = {int(c): r for r, c in enumerate(hk.tolist())}
            common = [c for c in ref["keep"].tolist() if c in pos]
            for a_, b_ in zip(common[:-1], common[1:]):          # adjacent pairs of the oracle's order that HIP reverses
                if pos[a_] > pos[b_]:
                    assert float(os_[a_] - os_[b_]) <= 2.0 * delta, (i, a_, b_)
        else:
            assert (det.boxes[i, :k].cpu() - ref["boxes"]).abs().max().item() < 2e-2
            assert (det.scores[i, :k].cpu() - ref["scores"]).abs().max().item() <= 3e-5
        s = det.scores[i, :k].cpu()
        assert (s[:-1] >= s[1:]).all()


def test_fcos_dropin_contract(fcos_sd, oracle_run):
    from fcos_utils.fcos import FCOS
    rgb, dets, _ = oracle_run
    model = FCOS(num_classes=3, ext=False, nms_thresh=0.5).cuda().eval()
    missing, unexpected = model.load_state_dict(fcos_sd, strict=False)
    assert not missing and not unexpected
    with torch.inference_mode():
        out = model([rgb[0].cuda(), rgb[1].cuda()], None)
    assert len(out) == 2
    d = out[0]
    assert set(d) == {"boxes", "scores", "labels", "sides", "feature_idx"}
    assert d["labels"].dtype == torch.int64 and d["sides"].dtype == torch.int64
    assert d["feature_idx"].dtype == torch.float32 and d["boxes"].shape[1] == 4 and d["boxes"].is_cuda
    assert abs(len(d["scores"]) - len(dets[0]["scores"])) <= 3


def test_fcos_dropin_honours_image_mean_std(fcos_sd):
    """FCOS(image_mean=..., image_std=...) forwards them to the transform (fcos.py:501-505): detections of the drop-in with
    a non-default normalisation equal the oracle's with the same one -- and differ from the default's."""
    from fcos_utils.fcos import FCOS
    from hn_amd import synth
    from oracle import fcos_ref
    mean, std = [0.40, 0.50, 0.45], [0.20, 0.25, 0.30]
    rgb = synth.make_rgb(2, seed=1234)
    ref = fcos_ref.fcos_forward([rgb[0], rgb[1]], fcos_sd, 3, image_mean=mean, image_std=std)
    base = fcos_ref.fcos_forward([rgb[0], rgb[1]], fcos_sd, 3)
    assert [len(d["keep"]) for d in ref] != [len(d["keep"]) for d in base] or \
        not torch.equal(ref[0]["boxes"], base[0]["boxes"])
    model = FCOS(num_classes=3, ext=False, image_mean=mean, image_std=std).cuda().eval()
    model.load_state_dict(fcos_sd, strict=False)
    with torch.inference_mode():
        out = model([rgb[0].cuda(), rgb[1].cuda()], None)
    for d, r in zip(out, ref):
        assert len(d["scores"]) == len(r["scores"])
        assert torch.equal(d["labels"].cpu(), r["labels"])
        assert (d["boxes"].cpu() - r["boxes"]).abs().max().item() < 2e-2
        assert (d["scores"].cpu() - r["scores"]).abs().max().item() < 1e-4


def test_fcos_ext_matches_reference_golden(golden_dir):
    """FCOS(ext=True) (class default, trainval_net_fcos.py --test-only) vs the imported reference:
    dict keys of fcos.py:637-647; per matched detection the contact state and side are identical and
    dxdymags agree to 1e-4 (logit noise of ~1e-5 may flip a detection sitting exactly on 0.7 / 0.3, hence a rate)."""
    from fcos_utils.fcos import FCOS
    from hn_amd import synth
    g = np.load(golden_dir / "fcos_ext_forward.npz")
    model = FCOS(num_classes=3).cuda().eval()
    assert model.ext
    missing, unexpected = model.load_state_dict(synth.make_fcos_state_dict(seed=0, num_classes=3, ext=True), strict=False)
    assert not missing and not unexpected
    rgb = synth.make_rgb(1, seed=int(g["rgb_seed"])).cuda()
    with torch.inference_mode():
        d = model([rgb[0]], None)[0]
    assert set(d) == {"boxes", "scores", "labels", "dxdymags", "contacts", "sides"}
    assert d["contacts"].dtype == torch.int64 and d["dxdymags"].shape[1] == 3
    boxes = d["boxes"].cpu()
    matched = 0
    for j, b in enumerate(torch.from_numpy(g["boxes"])):
        dist = (boxes - b).abs().max(dim=1)[0]
        i = int(dist.argmin())
        if dist[i] < 1e-2:
            matched += 1
            assert int(d["labels"][i]) == int(g["labels"][j]) and int(d["sides"][i]) == int(g["sides"][j])
            assert int(d["contacts"][i]) == int(g["contacts"][j])
            assert np.abs(d["dxdymags"][i].cpu().numpy() - g["dxdymags"][j]).max() < 1e-4
    assert matched >= 0.98 * len(g["labels"]) and len(boxes) <= 1.02 * len(g["labels"]) + 1


def test_head_side_streams_agree_with_default_path(fcos_sd, oracle_run):
    """FCOSEngine(head_streams=6) runs the six independent tower chains on side streams (off by default: it
    measured slower).  The default path groups the same convolutions into lock-step launches with other tile
    shapes, whose GroupNorm partial sums are reduced in a different order: same detections to fp32 rounding."""
    from hn_amd.fcos_engine import FCOSEngine
    rgb, _, _ = oracle_run
    a = FCOSEngine(fcos_sd, 3, device="cuda", head_streams=1)
    b = FCOSEngine(fcos_sd, 3, device="cuda", head_streams=6)
    da, _ = a.detect(rgb.cuda())
    db, _ = b.detect(rgb.cuda())
    torch.cuda.synchronize()
    for i in range(rgb.shape[0]):
        ka, kb = int(da.count[i]), int(db.count[i])
        assert ka > 0 and abs(ka - kb) <= 1
        k = min(ka, kb)
        same = ((da.boxes[i, :k] - db.boxes[i, :k]).abs().max(dim=1)[0] < 1e-3) & (da.labels[i, :k] == db.labels[i, :k])
        assert int(same.sum()) >= 0.98 * k


def test_other_frame_size_wide(fcos_sd, engine):
    """A 360x640 frame (16:9): the max_size branch of the resize rule (749x1333 -> canvas 768x1344, levels
    96x168 / 48x84 / 24x42) through the same kernels, vs the oracle; agreement as in the 480x640 end-to-end test."""
    from hn_amd import synth
    from oracle import fcos_ref
    rgb = synth.make_rgb(1, h=360, w=640, seed=77)
    dets, inter = fcos_ref.fcos_forward([rgb[0]], fcos_sd, 3, return_intermediates=True)
    oh, ow, ph, pw = engine.geometry(360, 640)
    assert (oh, ow) == tuple(inter["image_sizes"][0]) and (ph, pw) == tuple(inter["x"].shape[-2:]) == (768, 1344)
    det, cand = engine.detect(rgb.cuda())
    ref = dets[0]
    k = int(det.count[0])
    assert abs(int(cand.count[0]) - len(inter["candidates"][0]["scores"])) <= 3
    boxes, labels = det.boxes[0, :k].cpu(), det.labels[0, :k].cpu().long()
    matched = 0
    for b, l in zip(ref["boxes"], ref["labels"]):
        d = (boxes - b).abs().max(dim=1)[0]
        j = int(d.argmin())
        matched += int(d[j] < 1e-2 and labels[j] == l)
    assert len(ref["labels"]) > 20 and matched >= 0.98 * len(ref["labels"]) and k <= 1.02 * len(ref["labels"]) + 1


def test_mixed_size_image_list(fcos_sd):
    """torchvision batch_images (fcos_utils/fcos.py:702-709): a list of differently sized images is resized per
    image, padded to the common canvas and rescaled per image.  480x640 + 360x640 vs the oracle."""
    from fcos_utils.fcos import FCOS
    from hn_amd import ops, synth
    from hn_amd.fcos_engine import IMAGE_MEAN, IMAGE_STD
    from oracle import fcos_ref
    a = synth.make_rgb(1, seed=1000)[0]
    b = synth.make_rgb(1, seed=1001)[0][:, :360, :].contiguous()
    imgs = [a, b]
    dets, inter = fcos_ref.fcos_forward(imgs, fcos_sd, 3, return_intermediates=True)
    model = FCOS(num_classes=3, ext=False)
    model.load_state_dict(fcos_sd, strict=False)
    model = model.cuda().eval()
    eng = model.engine()
    geom, ph, pw = eng.list_geometry(imgs)
    assert [g[2:] for g in geom] == [tuple(s) for s in inter["image_sizes"]]
    assert (ph, pw) == tuple(inter["x"].shape[-2:])
    # staged: the canvas itself (fp32 mode of the list kernel) against the oracle's transform
    canvas = ops.fcos_preprocess_list([i.cuda() for i in imgs], geom, ph, pw, IMAGE_MEAN, IMAGE_STD, split=False)
    ref = inter["x"].permute(0, 2, 3, 1)
    assert (canvas[..., :3].cpu() - ref).abs().max().item() < 1e-5
    assert float(canvas[..., 3].abs().max()) == 0.0
    with torch.inference_mode():
        out = model([i.cuda() for i in imgs])
    assert len(out) == 2
    for got, want in zip(out, dets):
        k = want["scores"].numel()
        assert abs(got["scores"].numel() - k) <= max(1, k // 50)
        if got["scores"].numel() == k:
            same = (got["labels"].cpu() == want["labels"]) & ((got["boxes"].cpu() - want["boxes"]).abs().max(dim=1)[0] < 0.01)
            assert same.float().mean().item() >= 0.98
            assert (got["scores"].cpu() - want["scores"]).abs().max().item() < 1e-4


@pytest.mark.parametrize("k,classes", [(64, 1), (300, 3), (1200, 3), (3000, 2)])
def test_nms_kernel_satisfies_the_definition_of_greedy_nms(k, classes):
    """Row a8' anchor that does not involve the oracle: on tie-free boxes the survivors of hn_fcos_nms (coordinate-trick
    path up to 1000 candidates, per-class above; LDS sort up to 2048, global-memory sort above) are THE set the
    definition of greedy NMS admits -- tests/nms_property.py, IoUs recomputed in fp64 with a guard band around 0.3 --
    and hn_nms agrees on the single-class case."""
    import nms_property
    from hn_amd import ops
    boxes, scores, labels = nms_property.make_case(k, classes, seed=100 + k)
    cand = ops.alloc_candidates(1, k, "cuda")
    cand.boxes[0] = boxes.cuda()
    cand.scores[0] = scores.cuda()
    cand.labels[0] = labels.cuda()
    cand.count[0] = k
    det = ops.fcos_nms(cand, 0.3, 1.0, 1.0)
    n = int(det.count[0])
    keep = det.keep[0, :n].cpu().long().numpy()
    nms_property.check(boxes, scores, labels, keep, 0.3)
    if classes == 1:
        assert torch.equal(ops.nms(boxes.cuda(), scores.cuda(), 0.3).cpu(), torch.from_numpy(keep))


@pytest.mark.parametrize("n,h,w", [(2, 480, 640), (1, 96, 128), (3, 100, 260)])
def test_fused_stem_pool_equals_stem_then_pool(fcos_sd, n, h, w):
    """hn_conv_stem_pool_f16x3 (conv1 + bn1 + relu + 3x3/2 max pooling in one kernel, 15 x 17 conv patches per workgroup)
    is bit-identical to hn_conv_stem_f16x3 followed by hn_maxpool3x3s2_s32, including canvases whose pooled map is not a
    multiple of the 7 x 8 patch (partial patches at the bottom / right edge) and odd conv maps."""
    from hn_amd import ops, synth
    from hn_amd.fcos_engine import FCOSEngine, IMAGE_MEAN, IMAGE_STD
    eng = FCOSEngine(fcos_sd, 3, device="cuda")
    rgb = synth.make_rgb(n, seed=77, h=h, w=w).cuda() if "h" in synth.make_rgb.__code__.co_varnames else \
        torch.rand((n, 3, h, w), generator=torch.Generator().manual_seed(77)).cuda()
    oh, ow, ph, pw = eng.geometry(h, w)
    img16 = ops.fcos_preprocess_split(rgb, oh, ow, ph, pw, IMAGE_MEAN, IMAGE_STD)
    ref = ops.maxpool3x3s2_nhwc(ops.conv_stem_split(img16, eng.stem16.w16, eng.stem16.bias, 64, r=7, stride=2, relu=True))
    got = ops.conv_stem_pool_split(img16, eng.stem16.w16, eng.stem16.bias, 64, r=7, stride=2)
    assert got.shape == ref.shape
    assert torch.equal(got, ref)
    # and through the engine switch: same features either way
    a = eng.backbone(img16)
    eng.fuse_stem_pool = False
    b = eng.backbone(img16)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n,sizes,thresh", [
    (1, [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)], 0.7),
    (3, [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)], 0.5),      # ~half of the 17 850 points pass
    (2, [(9, 7), (3, 3)], 0.3),                                          # fewer points than one chunk
    (2, [(40, 26), (5, 5)], -1.0),                                       # every point passes: 1065 > one chunk, full capacity
])
def test_chunked_candidates_equal_the_single_workgroup_form(n, sizes, thresh):
    """hn_fcos_candidates_ws (count kernel + scatter kernel over 1024-point chunks) against hn_fcos_candidates (one
    workgroup per image): the same arithmetic from one shared device function and the same anchor order, so every output
    array and the counts must be identical (fcos.py:591-628)."""
    import ctypes as C
    from hn_amd import _lib, ops
    g = torch.Generator().manual_seed(99 + n)
    nc = 3
    cls = [torch.randn(n, h, w, nc + 2, generator=g).cuda() * 2 for h, w in sizes]
    reg = [torch.cat([torch.rand(n, h, w, 4, generator=g) * 3, torch.randn(n, h, w, 1, generator=g) * 2], -1).cuda() for h, w in sizes]
    strides = [8 * 2 ** i for i in range(len(sizes))]
    assert ops.CANDIDATES_CHUNKED
    a = ops.fcos_candidates(cls, reg, strides, nc, thresh)
    ops.CANDIDATES_CHUNKED = False
    try:
        b = ops.fcos_candidates(cls, reg, strides, nc, thresh)
    finally:
        ops.CANDIDATES_CHUNKED = True
    assert torch.equal(a.count, b.count) and int(a.count.min()) >= 0
    if thresh < 0:
        assert int(a.count.min()) == sum(h * w for h, w in sizes)
    for i in range(n):
        k = int(a.count[i])
        for f in ("boxes", "scores", "labels", "sides", "level", "point"):
            assert torch.equal(getattr(a, f)[i, :k], getattr(b, f)[i, :k]), f
    lib = _lib.load()
    assert lib.hn_fcos_candidates_ws_bytes(2, 17850) == 2 * 18 * 4 and lib.hn_fcos_candidates_ws_bytes(0, 5) == 0


@pytest.mark.parametrize("n,h,w", [(2, 480, 640), (1, 720, 1280), (3, 300, 333), (1, 1200, 1600)])
def test_tiled_preprocess_is_bit_identical_to_the_per_pixel_kernel(n, h, w, monkeypatch):
    """fcos_preprocess_split_tiled_kernel normalises the source rectangle of an 8 x 128 output tile once into LDS and
    interpolates from there: the same divisions on the same values and the same interpolation expressions as the per-pixel
    kernel (tv GeneralizedRCNNTransform: normalize, then bilinear resize, fcos.py:702-709) -> torch.equal on the stem image,
    for up- and down-scaling frames (the 1200 x 1600 frame is DOWNscaled: its tiles need more source rows than the patch
    holds only beyond scale 1.85, so it still takes the tiled kernel)."""
    from hn_amd import ops, synth
    from hn_amd.fcos_engine import FCOSEngine, IMAGE_MEAN, IMAGE_STD, resized_size
    x = synth.make_rgb(n, h, w, seed=7).cuda()
    oh, ow = resized_size(h, w, 800, 1333)
    ph, pw = (oh + 31) // 32 * 32, (ow + 31) // 32 * 32
    a = ops.fcos_preprocess_split(x, oh, ow, ph, pw, IMAGE_MEAN, IMAGE_STD)
    ops.set_form("preprocess_generic", True)
    try:
        b = ops.fcos_preprocess_split(x, oh, ow, ph, pw, IMAGE_MEAN, IMAGE_STD)
    finally:
        ops.set_form("preprocess_generic", False)
    assert a.shape == b.shape and torch.equal(a, b)

"""CPU tests: the oracle reproduces the golden vectors that the imported reference produced
(tests/golden/make_golden*.py), plus hand-checkable facts from SURVEY.md appendix A."""
import numpy as np
import pytest
import torch

from hn_amd import synth
from oracle import a2j_ref, fcos_ref, handnet_ref


def test_a2j_oracle_reproduces_reference_golden(golden_dir, a2j_sd):
    g = np.load(golden_dir / "a2j_forward.npz")
    x = synth.make_crops(2, 176, seed=int(g["input_seed"]))
    out, (x3, x4), raw = a2j_ref.a2j_forward(x, a2j_sd, return_heads=True)
    assert np.abs(out.numpy() - g["keypoints"]).max() <= 1e-5
    assert np.abs(x3[:, :64, 5, 5].numpy() - g["x3_probe"]).max() <= 1e-5
    assert np.abs(x4[:, :64, 5, 5].numpy() - g["x4_probe"]).max() <= 1e-5
    cls, reg, dep = a2j_ref.heads_to_reference_layout(*raw)
    assert np.abs(cls[:, :64].numpy() - g["cls_probe"]).max() <= 1e-5
    assert np.abs(reg[:, :64].numpy() - g["reg_probe"]).max() <= 1e-4
    assert np.abs(dep[:, :64].numpy() - g["dep_probe"]).max() <= 1e-5
    assert np.array_equal(a2j_ref.all_anchors().numpy(), g["anchors"])


def test_anchor_layout_facts():
    """first rows [2,2],[2,6],...,[14,14],[18,2]; coordinate 0 follows tensor dim H."""
    a = a2j_ref.all_anchors().numpy()
    assert a.shape == (1936, 2)
    assert a[0].tolist() == [2, 2] and a[1].tolist() == [2, 6] and a[15].tolist() == [14, 14]
    assert a[16].tolist() == [18, 2]
    # flat index n = (w*11 + h)*16 + a  <->  (h*16 + P[a//4], w*16 + P[a%4])
    P = [2, 6, 10, 14]
    for (w, h, k) in [(0, 0, 5), (3, 7, 9), (10, 10, 15)]:
        n = (w * 11 + h) * 16 + k
        assert a[n].tolist() == [h * 16 + P[k // 4], w * 16 + P[k % 4]]


def test_post_process_oracle_reproduces_reference_golden(golden_dir):
    g = np.load(golden_dir / "a2j_post_process.npz")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    cls = torch.randn((4, 1936, 21), generator=gen) * 2.0
    reg = torch.randn((4, 1936, 21, 2), generator=gen) * 8.0
    dep = 0.8 + 0.2 * torch.randn((4, 1936, 21), generator=gen)
    cls[3, 100, :] += 30.0
    cls[2] *= 0.0
    out = a2j_ref.post_process(cls, reg, dep)
    assert np.abs(out.numpy() - g["out"]).max() <= 1e-5


def test_transform_facts():
    """480x640 -> resize (800,1066) -> pad (800,1088); zeros outside.

    NOTE: SURVEY A.2 says 799 rows.  torchvision computes `self_min_size / min_size` with a
    python float on the left and a tensor on the right, i.e. Tensor.__rtruediv__ =
    reciprocal()*other: fp32(1/480)*800 = 1.66666674..., and floor(480*1.66666674) = 800.
    (799 comes from tensor(800.)/480., which is not what the transform evaluates.)"""
    img = synth.make_rgb(1, seed=5)[0]
    t, sizes = fcos_ref.transform([img])
    assert sizes == [(800, 1066)] and tuple(t.shape) == (1, 3, 800, 1088)
    assert float(t[0, :, :, 1066:].abs().max()) == 0.0
    assert fcos_ref.resized_size(480, 640) == (800, 1066)


def test_fcos_oracle_reproduces_reference_golden(golden_dir, fcos_sd):
    g = np.load(golden_dir / "fcos_forward.npz")
    rgb = synth.make_rgb(1, seed=int(g["rgb_seed"]))
    dets, inter = fcos_ref.fcos_forward([rgb[0]], fcos_sd, 3, return_intermediates=True)
    ho = inter["head"]
    assert inter["anchors"].shape[0] == int(g["num_anchors"]) == 17850
    assert np.array_equal(inter["anchors"][::97].numpy(), g["anchors_probe"])
    assert np.abs(ho["cls_logits"][0, ::97].numpy() - g["cls_probe"]).max() <= 1e-4
    assert np.abs(ho["bbox_regression"][0, ::97].numpy() - g["reg_probe"]).max() <= 1e-4
    assert np.abs(ho["bbox_ctrness"][0, ::97].numpy() - g["ctr_probe"]).max() <= 1e-4
    assert np.abs(ho["hand_lr"][0, ::97].numpy() - g["lr_probe"]).max() <= 1e-4
    d = dets[0]
    assert len(inter["candidates"][0]["scores"]) == int(g["n_candidates"])
    assert np.array_equal(d["labels"].numpy(), g["labels"])
    assert np.array_equal(d["sides"].numpy(), g["sides"])
    assert np.array_equal(d["feature_idx"].numpy(), g["feature_idx"])
    assert np.abs(d["boxes"].numpy() - g["boxes"]).max() <= 1e-3
    assert np.abs(d["scores"].numpy() - g["scores"]).max() <= 1e-6
    assert (np.diff(d["scores"].numpy()) <= 0).all()  # score-descending


def test_fcos_ext_oracle_reproduces_reference_golden(golden_dir):
    """ext=True heads (fcos.py:255-264,299-320) and dict (fcos.py:637-647)."""
    g = np.load(golden_dir / "fcos_ext_forward.npz")
    sd = synth.make_fcos_state_dict(seed=0, num_classes=3, ext=True)
    rgb = synth.make_rgb(1, seed=int(g["rgb_seed"]))
    d = fcos_ref.fcos_forward([rgb[0]], sd, 3, ext=True)[0]
    assert np.array_equal(d["labels"].numpy(), g["labels"])
    assert np.array_equal(d["sides"].numpy(), g["sides"])
    assert np.array_equal(d["contacts"].numpy(), g["contacts"])
    assert np.abs(d["dxdymags"].numpy() - g["dxdymags"]).max() <= 1e-5
    assert np.abs(d["boxes"].numpy() - g["boxes"]).max() <= 1e-3
    assert np.abs(d["scores"].numpy() - g["scores"]).max() <= 1e-6


def test_resnet34_trunk_matches_intree_basicblock_resnet(golden_dir, fcos_sd):
    """Stem + layer1-3 of the restated torchvision ResNet-34 trunk vs the reference's in-tree ResNet(BasicBlock)."""
    g = np.load(golden_dir / "resnet34_intree.npz")
    x = torch.randn((1, 3, 96, 128), generator=torch.Generator().manual_seed(int(g["input_seed"])))
    c2, c3, c4, _ = fcos_ref.body(x, fcos_sd)
    assert np.abs(c2[:, ::8].numpy() - g["c2"]).max() <= 1e-5
    assert np.abs(c3[:, ::16].numpy() - g["c3"]).max() <= 1e-5
    assert np.abs(c4[:, ::32].numpy() - g["c4"]).max() <= 1e-5


def test_handnet_oracle_reproduces_reference_golden(golden_dir, fcos_sd, a2j_sd):
    g = np.load(golden_dir / "handnet_forward.npz")
    rgb = synth.make_rgb(2, seed=int(g["rgb_seed"]))
    depth = synth.make_depth(2, seed=int(g["depth_seed"]))
    kp, depth_batch, crops = handnet_ref.handnet_forward([rgb[0], rgb[1]], depth, fcos_sd, a2j_sd, 3)
    assert crops.dtype == torch.int64 and np.array_equal(crops.numpy(), g["crops"])
    assert np.array_equal(depth_batch[:, 0, ::16, ::16].numpy(), g["depth_batch_probe"])
    assert abs(depth_batch.double().sum().item() - float(g["depth_batch_sum"])) < 1e-6
    assert np.abs(kp.numpy() - g["keypoints"]).max() <= 1e-4


def test_rgbd_oracles_reproduce_reference_goldens(golden_dir, fcos_sd, a2j_rgbd_sd):
    """4-channel stem (a2j/a2j.py:191-199) and the RGBD crop permutation (handnet_pipeline.py:102)."""
    g = np.load(golden_dir / "a2j_rgbd_forward.npz")
    x = synth.make_rgbd_crops(2, 176, seed=int(g["input_seed"]))
    out, (x3, x4), _ = a2j_ref.a2j_forward(x, a2j_rgbd_sd, channel_in=4, return_heads=True)
    assert np.abs(out.numpy() - g["keypoints"]).max() <= 1e-5
    assert np.abs(x3[:, :64, 5, 5].numpy() - g["x3_probe"]).max() <= 1e-5
    assert np.abs(x4[:, :64, 5, 5].numpy() - g["x4_probe"]).max() <= 1e-5
    g = np.load(golden_dir / "handnet_rgbd_forward.npz")
    rgb = synth.make_rgb(2, seed=int(g["rgb_seed"]))
    depth = synth.make_depth(2, seed=int(g["depth_seed"]))
    kp, depth_batch, crops = handnet_ref.handnet_forward([rgb[0], rgb[1]], torch.cat([rgb, depth], 1), fcos_sd,
                                                         a2j_rgbd_sd, 3, rgbd=True)
    assert np.array_equal(crops.numpy(), g["crops"])
    assert np.array_equal(depth_batch[:, :, ::16, ::16].numpy(), g["depth_batch_probe"])
    assert np.abs(depth_batch.double().sum(dim=(0, 2, 3)).numpy() - g["depth_batch_sum"]).max() < 1e-6
    assert np.abs(kp.numpy() - g["keypoints"]).max() <= 1e-4


def test_crop_box_rule():
    """trunc, pad 0.4, clamp to [0,W]/[0,H]."""
    b = handnet_ref.crop_box(torch.tensor([[10.9, 20.2, 110.7, 220.9]]), 640, 480)
    assert b.tolist() == [0, 0, 150, 300]           # 10-40 <0 -> 0 ; 20-80 -> 0 ; 110+40 ; 220+80
    b = handnet_ref.crop_box(torch.tensor([[600.0, 400.0, 639.9, 479.9]]), 640, 480)
    assert b.tolist() == [584, 368, 640, 480]
    d = torch.arange(480 * 640, dtype=torch.float32).reshape(1, 480, 640)
    c = handnet_ref.crop_depth(d, b)
    assert tuple(c.shape) == (1, 176, 176) and c[0, 0, 0].item() == d[0, 368, 584].item()
    assert c[0, -1, -1].item() == d[0, 479, 639].item()  # inclusive slice clamps at H/W


def test_nms_oracle_semantics():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.9])
    keep = fcos_ref.nms(boxes, scores, 0.3)
    assert keep.tolist() == [0, 2]                  # tie 0/3 -> lower index first; 1 and 3 suppressed
    assert fcos_ref.nms(boxes[:0], scores[:0], 0.3).tolist() == []
    # boundary: IoU exactly float32(0.3) IS suppressed (fp32 IoU compared with double 0.3)
    b2 = torch.tensor([[0, 0, 10, 13], [0, 0, 10, 3.0]])  # inter 30 / union 130... use exact 0.3 case below
    b2 = torch.tensor([[0, 0, 10, 10], [0, 0, 10, 3.0]])  # inter 30 / union 100 = 0.3
    assert fcos_ref.nms(b2, torch.tensor([0.9, 0.8]), 0.3).tolist() == [0]
    # batched: different classes never suppress each other (0 and 3 are identical boxes of
    # classes 0 / 1 and both survive); 1 (class 1) is suppressed by 3 (class 1)
    k = fcos_ref.batched_nms(boxes, scores, torch.tensor([0, 1, 0, 1]), 0.3)
    assert k.tolist() == [0, 3, 2]


def test_pose2mesh_oracle_reproduces_reference_golden(golden_dir):
    """FlatPose2Mesh (PoseNet MLP + Chebyshev graph-conv mesh net) restated vs the imported reference."""
    from oracle import pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    sd = synth.make_pose2mesh_state_dict(seed=int(g["weight_seed"]), graph_sizes=[m.shape[0] for m in graphs])
    pose2d = torch.randn((3, 21, 2), generator=torch.Generator().manual_seed(int(g["input_seed"])))
    mesh, pose3d = pose2mesh_ref.pose2mesh_forward(pose2d, sd, graphs)
    assert mesh.shape == (3, graphs[0].shape[0], 3) and pose3d.shape == (3, 21, 3)
    assert np.abs(pose3d.numpy() - g["pose3d"]).max() <= 1e-3          # values ~1e2 (millimetres)
    assert np.abs(mesh.numpy() - g["mesh"]).max() <= 1e-4


def test_frozen_bn_matches_published_copy_of_torchvision_class(fcos_sd):
    """torchvision itself is absent from the image, but HF transformers ships a verbatim copy of its
    FrozenBatchNorm2d ("Copy-paste from torchvision.misc.ops", transformers/models/detr/modeling_detr.py) with the
    eps of torchvision >= 0.5: the oracle's _frozen_bn (ResNet-34 body of the FCOS detector, fcos.py:476,737) must
    reproduce it bit for bit on the synthetic checkpoint's statistics."""
    detr = pytest.importorskip("transformers.models.detr.modeling_detr")
    name = "backbone.body.layer2.0.bn1"
    c = fcos_sd[name + ".weight"].numel()
    m = detr.DetrFrozenBatchNorm2d(c)
    with torch.no_grad():
        for k in ("weight", "bias", "running_mean", "running_var"):
            getattr(m, k).copy_(fcos_sd[f"{name}.{k}"])
    x = torch.randn((2, c, 9, 7), generator=torch.Generator().manual_seed(4))
    assert torch.equal(fcos_ref._frozen_bn(x, fcos_sd, name), m(x))


def test_resnet34_body_matches_third_party_basic_block_resnet(fcos_sd):
    """Row a4, third-party pin for the whole trunk INCLUDING layer4 (torchvision is absent from the image): Hugging
    Face's ResNetModel in its "basic" configuration (depths 3-4-6-3, stride on the first 3x3 of a stage, 1x1 + BN
    shortcut, 7x7/2 stem + 3x3/2 max-pool) is the torchvision resnet34 architecture that fcos.py:737 builds.  With the
    synthetic checkpoint's weights copied in, its four stage outputs must equal the oracle's C2..C5 (eval-mode
    BatchNorm vs the folded FrozenBN form: a few ulps per layer)."""
    tr = pytest.importorskip("transformers")
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[3, 4, 6, 3],
                          layer_type="basic", hidden_act="relu", downsample_in_first_stage=False)
    m = tr.ResNetModel(cfg).eval()
    p = "backbone.body."
    new = {}

    def bn(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var"):
            new[f"{dst}.{k}"] = fcos_sd[f"{src}.{k}"]
    new["embedder.embedder.convolution.weight"] = fcos_sd[p + "conv1.weight"]
    bn("embedder.embedder.normalization", p + "bn1")
    for li, blocks in enumerate((3, 4, 6, 3), start=1):
        for b in range(blocks):
            src, dst = f"{p}layer{li}.{b}.", f"encoder.stages.{li - 1}.layers.{b}."
            for i in (1, 2):
                new[f"{dst}layer.{i - 1}.convolution.weight"] = fcos_sd[f"{src}conv{i}.weight"]
                bn(f"{dst}layer.{i - 1}.normalization", f"{src}bn{i}")
            if f"{src}downsample.0.weight" in fcos_sd:
                new[f"{dst}shortcut.convolution.weight"] = fcos_sd[f"{src}downsample.0.weight"]
                bn(f"{dst}shortcut.normalization", f"{src}downsample.1")
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    x = torch.randn((2, 3, 96, 128), generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        theirs = m(x, output_hidden_states=True).hidden_states[1:]
        ours = fcos_ref.body(x, fcos_sd)
    assert len(theirs) == 4
    for name, a, b in zip(("C2", "C3", "C4", "C5"), ours, theirs):
        assert a.shape == b.shape, name
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-5 * max(scale, 1.0), name


@pytest.mark.parametrize("k,classes", [(64, 1), (300, 3), (1200, 3), (3000, 2)])
def test_nms_oracle_satisfies_the_definition_of_greedy_nms(k, classes):
    """Row a8' anchor that needs neither torchvision nor the HIP kernel: on tie-free boxes the kept set of the C
    restatement (both batched_nms paths: coordinate trick up to 1000 boxes, per-class above) is THE set the definition
    of greedy NMS admits (tests/nms_property.py), IoUs recomputed in fp64 with a guard band around 0.3."""
    import nms_property
    boxes, scores, labels = nms_property.make_case(k, classes, seed=100 + k)
    keep = fcos_ref.batched_nms(boxes, scores, labels.to(torch.int64), 0.3)
    nms_property.check(boxes, scores, labels, keep.numpy(), 0.3)
    if classes == 1:
        keep1 = fcos_ref.nms(boxes, scores, 0.3)
        assert torch.equal(keep1, keep)


def test_fpn_top_down_matches_hand_written_fp64_loops(fcos_sd):
    """Row a4 (FPN wiring, torchvision 0.11.3 ops/feature_pyramid_network.py, call site fcos.py:476): the oracle's
    top-down pathway -- nearest-neighbour 2x upsampling of the coarser lateral, added to the finer lateral BEFORE the
    3x3 output conv, coarsest level first -- against explicit fp64 loops written from the definition
    (src index = floor(dst * in / out)), on odd map sizes where 'nearest' and 'size=' matter."""
    g = torch.Generator().manual_seed(5)
    cs = [torch.randn((1, 128, 13, 18), generator=g), torch.randn((1, 256, 7, 9), generator=g),
          torch.randn((1, 512, 4, 5), generator=g)]
    f = "backbone.fpn."
    lat = [torch.nn.functional.conv2d(c.double(), fcos_sd[f"{f}inner_blocks.{i}.weight"].double(),
                                      fcos_sd[f"{f}inner_blocks.{i}.bias"].double()) for i, c in enumerate(cs)]
    merged = [None, None, lat[2]]
    for lvl in (1, 0):
        coarse, fine = merged[lvl + 1], lat[lvl]
        _, ch, fh, fw = fine.shape
        ih, iw = coarse.shape[-2:]
        out = fine.clone()
        for y in range(fh):
            sy = min(int(np.floor(y * (ih / fh))), ih - 1)
            for x in range(fw):
                sx = min(int(np.floor(x * (iw / fw))), iw - 1)
                out[0, :, y, x] += coarse[0, :, sy, sx]
        merged[lvl] = out
    want = [torch.nn.functional.conv2d(m, fcos_sd[f"{f}layer_blocks.{i}.weight"].double(),
                                       fcos_sd[f"{f}layer_blocks.{i}.bias"].double(), padding=1) for i, m in enumerate(merged)]
    # the oracle's FPN on the same C3..C5 (fcos_ref.backbone takes an image; its FPN half is restated here call by call)
    import oracle.fcos_ref as R
    got = R.fpn(cs, fcos_sd)
    for i in range(3):
        assert got[i].shape == want[i].shape
        assert (got[i].double() - want[i]).abs().max().item() < 1e-4 * max(1.0, want[i].abs().max().item())


def test_lifter_input_chain_is_a_per_axis_standardisation():
    """oracle.pose2mesh_ref.lifter_input restates ros_demo.py:148-157 function by function (get_bbox, process_bbox,
    j2d_processing's affine map at rot 0, / input_shape, (x - mean) / std).  With no rotation that chain is a per-axis positive
    affine map followed by a per-axis standardisation, which cancels the map: it equals (x - mean) / std of the image joints
    to the chain's own fp32 rounding -- what hn_joints2d_standardize_f32 computes on the device."""
    import numpy as np
    from oracle import pose2mesh_ref
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(100):
        j = (rng.uniform(50, 600, 2) + rng.normal(size=(21, 2)) * rng.uniform(5, 150, 2)).astype(np.float32)
        a = pose2mesh_ref.lifter_input(j)
        j64 = j.astype(np.float64)
        worst = max(worst, float(np.abs(a - (j64 - j64.mean(0)) / j64.std(0)).max()))
    assert worst < 3e-5
    assert pose2mesh_ref.lifter_input(np.full((21, 2), 7.0, dtype=np.float32)) is None    # degenerate box: the caller skips the frame
    # the affine map really is isotropic scale + translation (three-point solve, as cv2.getAffineTransform)
    t = pose2mesh_ref._affine_transform_rot0(np.array([100.0, 50.0], np.float32), np.array([40.0, 30.0], np.float32), (288, 384))
    assert abs(t[0, 1]) < 1e-9 and abs(t[1, 0]) < 1e-9 and abs(t[0, 0] - t[1, 1]) < 1e-9 and t[0, 0] > 0


def test_lifter_input_pieces_match_the_imported_reference(golden_dir):
    """oracle.pose2mesh_ref.get_bbox / process_bbox / the centre-scale step / the affine point map against outputs of the
    reference's own coord_utils.py and aug_utils.py (tests/golden/make_golden_lifter_input.py IMPORTS them): everything of
    ros_demo.py:148-157 except the one cv2.getAffineTransform call (OpenCV is absent: restated as the three-point solve)."""
    import numpy as np
    from oracle import pose2mesh_ref as r
    g = np.load(golden_dir / "lifter_input.npz")
    assert tuple(g["input_shape"]) == r.INPUT_SHAPE
    for i in range(g["joints"].shape[0]):
        b = r.get_bbox(g["joints"][i])
        assert np.array_equal(b, g["bbox"][i])
        b2 = r.process_bbox(b.copy())
        assert (b2 is not None) == bool(g["ok"][i])
        if b2 is None:
            assert r.lifter_input(g["joints"][i]) is None
            continue
        assert np.array_equal(np.asarray(b2, dtype=np.float64), g["bbox2"][i])
        center = np.array([b2[0] + b2[2] * 0.5, b2[1] + b2[3] * 0.5], dtype=np.float32)
        scale = np.array([b2[2] * 1.0, b2[3] * 1.0], dtype=np.float32)
        assert np.array_equal(center, g["center"][i]) and np.array_equal(scale, g["scale"][i])
    for t, ps, ws in zip(g["t"], g["pts"], g["warped"]):
        for p, w in zip(ps, ws):
            assert np.array_equal(np.dot(t, np.array([p[0], p[1], 1.]).T)[:2], w)       # affine_transform, aug_utils.py:176-179


def test_live_parity_harness_on_the_oracles_own_chain(golden_dir, fcos_sd, a2j_sd):
    """oracle.parity.live_parity (bench.py's other_configs.live_b1.parity): fed the oracle chain's own outputs as "the HIP
    step's record" it reports zero differences on an identical crop box; with the box moved by one pixel the frame is not
    compared (a different integer crop is a different input); a mesh moved by 1e-2 is outside the tolerance."""
    from oracle import parity, pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    p2m_sd = synth.make_pose2mesh_state_dict(0, graph_sizes=[m.shape[0] for m in graphs])
    paras = (617.343, 617.343, 312.42, 241.42)
    rgb, depth = synth.make_rgb(1, seed=1000), synth.make_depth(1, seed=2000)
    kp, _d, crops = handnet_ref.handnet_forward([rgb[0]], depth, fcos_sd, a2j_sd, 3)
    det = crops[0].clone()
    det[:2], det[2:] = torch.clamp(det[:2], 0, 480), torch.clamp(det[2:], 0, 640)
    k = torch.clamp(kp[0], 0.0, 176.0).numpy()
    j2d, j3d = a2j_ref.convert_joints(k, det.numpy(), None), a2j_ref.convert_joints(k, det.numpy(), paras)
    mesh, _ = pose2mesh_ref.pose2mesh_forward(torch.from_numpy(pose2mesh_ref.lifter_input(j2d[:, :2]))[None], p2m_sd, graphs)
    hip = (kp, crops, torch.from_numpy(j2d)[None], torch.from_numpy(j3d)[None], mesh)
    stats, whole_s, lift_s = parity.live_parity(hip, rgb, depth, fcos_sd, a2j_sd, p2m_sd, graphs, paras, reps=1)
    assert stats["frames"] == 1 and stats["crop_box_identical"] == 1 and stats["mesh_within_tolerance"] is True
    assert stats["max_abs_keypoint_diff"] == 0 and stats["max_abs_xyz_diff_mm"] == 0 and stats["max_abs_mesh_vertex_diff"] == 0
    assert 0 < lift_s < whole_s
    moved = (kp, crops + 1, hip[2], hip[3], mesh)
    stats, _, _ = parity.live_parity(moved, rgb, depth, fcos_sd, a2j_sd, p2m_sd, graphs, paras, reps=1)
    assert stats["crop_box_identical"] == 0 and stats["mesh_within_tolerance"] is False
    off = (kp, crops, hip[2], hip[3], mesh + 1e-2)
    stats, _, _ = parity.live_parity(off, rgb, depth, fcos_sd, a2j_sd, p2m_sd, graphs, paras, reps=1)
    assert stats["mesh_within_tolerance"] is False and abs(stats["max_abs_mesh_vertex_diff"] - 1e-2) < 1e-6

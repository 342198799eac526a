"""Crop stage and the end-to-end HandNet callable on HIP vs the oracle / reference goldens."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dets_from_boxes(boxes, labels):
    """Pack per-image (boxes [k,4], labels [k]) lists into an ops.Detections."""
    from hn_amd import ops
    n = len(boxes)
    cap = max(8, max(len(b) for b in boxes))
    det = ops.alloc_detections(n, cap, "cuda")
    for i, (b, l) in enumerate(zip(boxes, labels)):
        k = len(b)
        if k:
            det.boxes[i, :k] = torch.as_tensor(b, dtype=torch.float32).cuda()
            det.labels[i, :k] = torch.as_tensor(l, dtype=torch.int32).cuda()
        det.count[i] = k
    return det


def test_crop_resize_matches_reference_rule():
    """Boxes incl. negative / out-of-range / tiny / full-frame; hand = first label-2 box in score order."""
    from hn_amd import ops
    from oracle import handnet_ref
    g = torch.Generator().manual_seed(9)
    depth = 0.3 + torch.rand((6, 1, 480, 640), generator=g)
    boxes = [
        [[100.7, 50.2, 300.9, 400.5]],
        [[5.0, 5.0, 9.0, 9.0], [-30.5, 200.2, 40.9, 260.0]],       # first is label 0; second (negative x1) is the hand
        [[600.1, 440.3, 700.0, 500.0]],                              # sticks out bottom/right
        [[0.0, 0.0, 639.9, 479.9]],                                  # full frame
        [[10.0, 10.0, 11.5, 11.2]],                                  # 1x1 px box
        [[50.0, 60.0, 70.0, 80.0]],                                  # no hand label at all
    ]
    labels = [[2], [0, 2], [2], [2], [2], [1]]
    det = _dets_from_boxes(boxes, labels)
    crop_box, has_hand, crops = ops.crop_resize(det, 2, depth.cuda(), 176, 4)
    assert has_hand.cpu().tolist() == [1, 1, 1, 1, 1, 0]
    for i in range(6):
        if not has_hand[i]:
            assert crop_box[i].cpu().tolist() == [0, 0, 0, 0] and float(crops[i].abs().max()) == 0.0
            continue
        hb = torch.tensor([b for b, l in zip(boxes[i], labels[i]) if l == 2][:1])
        ref_box = handnet_ref.crop_box(hb, 640, 480)
        assert crop_box[i].cpu().tolist() == ref_box.tolist()          # int64, bit-exact
        ref_crop = handnet_ref.crop_depth(depth[i], ref_box)
        assert torch.equal(crops[i, :, :, 0].cpu(), ref_crop[0])         # pure gather, bit-exact
        assert float(crops[i, :, :, 1:].abs().max()) == 0.0


def test_crop_degenerate_box_counts_as_no_hand():
    from hn_amd import ops
    depth = torch.ones((1, 1, 480, 640))
    det = _dets_from_boxes([[[700.0, 100.0, 760.0, 200.0]]], [[2]])  # entirely right of the image
    crop_box, has_hand, crops = ops.crop_resize(det, 2, depth.cuda(), 176, 4)
    assert has_hand.cpu().tolist() == [0] and float(crops.abs().max()) == 0.0


@pytest.fixture(scope="module")
def handnet(fcos_sd, a2j_sd):
    from handnet_pipeline.handnet_pipeline import HandNet
    args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j="unused.pth")
    net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=False, RGBD=False)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    return net.cuda().eval()


def test_handnet_matches_reference_golden(handnet, golden_dir):
    """Same call as ros_demo.py:270; compared with what the imported reference returned."""
    from hn_amd import synth
    g = np.load(golden_dir / "handnet_forward.npz")
    rgb = synth.make_rgb(2, seed=int(g["rgb_seed"])).cuda()
    depth = synth.make_depth(2, seed=int(g["depth_seed"])).cuda()
    with torch.inference_mode():
        kp, depth_batch, crops = handnet([rgb[0], rgb[1]], depth_images=depth)
    assert kp.device.type == "cpu" and kp.shape == (2, 21, 3) and kp.dtype == torch.float32
    assert crops.dtype == torch.int64 and crops.is_cuda and depth_batch.is_cuda
    assert depth_batch.shape == (2, 1, 176, 176)
    assert np.array_equal(crops.cpu().numpy(), g["crops"])                       # int boxes bit-exact
    assert np.array_equal(depth_batch[:, 0, ::16, ::16].cpu().numpy(), g["depth_batch_probe"])
    assert abs(depth_batch.double().sum().item() - float(g["depth_batch_sum"])) < 1e-6
    assert np.abs(kp.numpy() - g["keypoints"]).max() < 1e-3                      # north_star tolerance


def test_handnet_contract_branches(handnet):
    from handnet_pipeline import HandNetPipeline
    from hn_amd import synth
    assert HandNetPipeline is type(handnet)
    rgb = synth.make_rgb(1, seed=5).cuda()
    depth = synth.make_depth(1, seed=6).cuda()
    assert handnet([rgb[0]], depth_images=depth, is_detect=True) is None
    assert handnet([rgb[0]], depth_images=depth, is_3D=True) is None
    # no detection: a black frame scores below 0.7 everywhere?  force it by demanding a class that never wins
    old = handnet.num_classes
    handnet._invalidate_all()
    handnet.num_classes = 99
    try:
        kp, db, cr = handnet([rgb[0]], depth_images=depth)
    finally:
        handnet.num_classes = old
        handnet._invalidate_all()
    assert kp.shape == (1, 21, 3) and float(kp.abs().max()) == 0.0
    assert db.shape == depth.shape and float(db.abs().max()) == 0.0
    assert cr.shape == (1, 4) and cr.dtype == torch.float32


def test_pipeline_batch_consistency_and_graph(handnet):
    """Frames are independent: a batch-8 run equals per-frame runs (the property the multi-GPU
    sharding relies on), and a hipGraph replay equals the eager run."""
    from hn_amd import synth
    rgb = synth.make_rgb(8, seed=1000).cuda()
    depth = synth.make_depth(8, seed=2000).cuda()
    eng = handnet.engine()
    full = eng.forward_device(rgb, depth)
    assert int(full.has_hand.sum()) >= 6
    for i in (0, 5):
        one = eng.forward_device(rgb[i:i + 1], depth[i:i + 1])
        assert torch.equal(one.crop_box[0], full.crop_box[i])
        assert (one.keypoints[0] - full.keypoints[i]).abs().max().item() < 1e-4
    run, s_img, s_dep, out = eng.graphed(rgb, depth)
    s_img.copy_(rgb)
    s_dep.copy_(depth)
    run()
    torch.cuda.synchronize()
    assert torch.equal(out.crop_box, full.crop_box)
    # graph capture also splits short k loops (launches are free there): same values to fp32 rounding, not bitwise
    assert (out.keypoints - full.keypoints).abs().max().item() < 3e-4   # ~1e-6 relative on pixel coordinates ~1e2


def test_rgbd_crop_reorders_channels():
    """handnet_pipeline.py:101-102: 4-channel crop, output channels = input channels [2,1,0,3]."""
    from hn_amd import ops
    from oracle import handnet_ref
    g = torch.Generator().manual_seed(11)
    rgbd = torch.rand((2, 4, 480, 640), generator=g)
    boxes = [[[100.7, 50.2, 300.9, 400.5]], [[600.1, 440.3, 700.0, 500.0]]]
    det = _dets_from_boxes(boxes, [[2], [2]])
    crop_box, has_hand, crops = ops.crop_resize(det, 2, rgbd.cuda(), 176, 4, reorder_bgr=True)
    for i in range(2):
        ref_box = handnet_ref.crop_box(torch.tensor(boxes[i]), 640, 480)
        ref = handnet_ref.crop_depth(rgbd[i], ref_box)[[2, 1, 0, 3]]
        assert crop_box[i].cpu().tolist() == ref_box.tolist()
        assert torch.equal(crops[i].permute(2, 0, 1).cpu(), ref)
    _, _, plain = ops.crop_resize(det, 2, rgbd.cuda(), 176, 4, reorder_bgr=False)
    assert torch.equal(plain[..., [2, 1, 0, 3]], crops)


def test_handnet_rgbd_matches_reference_golden(fcos_sd, a2j_rgbd_sd, golden_dir, tmp_path):
    """RGBD=True through the drop-in, loading A2J from a Lightning-style .ckpt (handnet_pipeline.py:28-29:
    weights under state_dict['a2j.*'])."""
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    ckpt = tmp_path / "a2j_rgbd.ckpt"
    torch.save({"state_dict": {"a2j." + k: v for k, v in a2j_rgbd_sd.items()},
                "hyper_parameters": {"num_classes": 21, "is_RGBD": True}}, ckpt)
    args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j=str(ckpt))
    net = HandNet(args, reload_detector=False, num_classes=3, reload_a2j=True, RGBD=True)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net = net.cuda().eval()
    g = np.load(golden_dir / "handnet_rgbd_forward.npz")
    rgb = synth.make_rgb(2, seed=int(g["rgb_seed"])).cuda()
    depth = synth.make_depth(2, seed=int(g["depth_seed"])).cuda()
    with torch.inference_mode():
        kp, depth_batch, crops = net([rgb[0], rgb[1]], depth_images=torch.cat([rgb, depth], 1))
    assert depth_batch.shape == (2, 4, 176, 176)
    assert np.array_equal(crops.cpu().numpy(), g["crops"])
    assert np.array_equal(depth_batch[:, :, ::16, ::16].cpu().numpy(), g["depth_batch_probe"])
    assert np.abs(kp.numpy() - g["keypoints"]).max() < 1e-3
    with pytest.raises(ValueError):
        net([rgb[0], rgb[1]], depth_images=depth)      # an RGBD model needs the 4-channel tensor


def test_handnet_graph_mode_equals_eager(handnet):
    """HandNet.enable_graph(): captured replay returns what the eager call returns, also for new inputs of the
    same shape (inputs are copied into the captured buffers)."""
    from hn_amd import synth
    depth = synth.make_depth(2, seed=31).cuda()
    outs = {}
    for mode in (False, True):
        handnet.enable_graph(mode)
        for seed in (41, 42):
            rgb = synth.make_rgb(2, seed=seed).cuda()
            with torch.inference_mode():
                kp, db, cr = handnet([rgb[0], rgb[1]], depth_images=depth)
            outs[(mode, seed)] = (kp.clone(), db.clone(), cr.clone())
    handnet.enable_graph(False)
    for seed in (41, 42):
        for a, b in zip(outs[(False, seed)], outs[(True, seed)]):
            assert a.shape == b.shape and (a.float() - b.float()).abs().max().item() < 3e-4
    assert not torch.equal(outs[(True, 41)][0], outs[(True, 42)][0])


def test_convert_joints_matches_reference_formula():
    """crop-uvd -> image-uvd -> camera xyz (mm) on the device vs the oracle's restatement of
    a2j/a2j.py:17-34 + datasets3d/a2jdataset.py:31-38.  Tolerance 1e-2 mm on O(100-1000) mm values."""
    from hn_amd import ops
    from oracle import a2j_ref
    g = torch.Generator().manual_seed(3)
    kp = torch.rand((5, 21, 3), generator=g) * torch.tensor([176.0, 176.0, 1.2]) + torch.tensor([0.0, 0.0, 0.3])
    box = torch.tensor([[0, 200, 49, 266], [192, 0, 264, 46], [100, 50, 420, 430], [0, 0, 640, 480], [7, 9, 8, 10]])
    valid = torch.tensor([1, 1, 0, 1, 1], dtype=torch.int32)
    paras = (617.343, 617.343, 312.42, 241.42)
    xyz = ops.convert_joints(kp.cuda(), box.cuda(), valid.cuda(), paras).cpu()
    uvd = ops.convert_joints(kp.cuda(), box.cuda(), None, None).cpu()
    for i in range(5):
        ref_xyz = a2j_ref.convert_joints(kp[i].numpy(), box[i].numpy(), paras)
        ref_uvd = a2j_ref.convert_joints(kp[i].numpy(), box[i].numpy(), None)
        assert np.abs(uvd[i].numpy() - ref_uvd).max() < 1e-3
        if valid[i]:
            assert np.abs(xyz[i].numpy() - ref_xyz).max() < 1e-2
        else:
            assert float(xyz[i].abs().max()) == 0.0


def test_pipeline_batch32_permutation_invariance(handnet):
    """BASELINE config 4 size: frames are independent, so permuting the 32 frames of a batch permutes the results
    (what the contiguous multi-GPU sharding relies on) -- integer boxes exactly, keypoints to fp32 rounding."""
    from hn_amd import synth
    rgb = synth.make_rgb(32, seed=1234).cuda()
    depth = synth.make_depth(32, seed=4321).cuda()
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(5)).cuda()
    eng = handnet.engine()
    a = eng.forward_device(rgb, depth)
    b = eng.forward_device(rgb[perm].contiguous(), depth[perm].contiguous())
    torch.cuda.synchronize()
    assert int(a.has_hand.sum()) >= 24
    assert torch.equal(a.has_hand[perm], b.has_hand) and torch.equal(a.crop_box[perm], b.crop_box)
    assert torch.equal(a.detections.count[perm], b.detections.count)
    assert (a.keypoints[perm] - b.keypoints).abs().max().item() < 1e-4


def test_convert_joints_matches_reference_golden(golden_dir):
    """hn_convert_joints_f32 against outputs of the reference's own a2j.a2j.convert_joints + uvd2xyz
    (tests/golden/make_golden_joints.py imports them): camera xyz in mm and the paras=None image-uvd branch."""
    from hn_amd import ops
    g = np.load(golden_dir / "convert_joints.npz")
    kp, box = torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["box"]).cuda()
    uvd = ops.convert_joints(kp, box, None, None).cpu().numpy()
    assert np.abs(uvd - g["uvd_img"]).max() < 2e-4            # pixels (|u| <= 640)
    for i in range(kp.shape[0]):
        xyz = ops.convert_joints(kp[i:i + 1], box[i:i + 1], None, tuple(float(v) for v in g["paras"][i])).cpu().numpy()[0]
        assert np.abs(xyz - g["xyz_pred"][i]).max() < 1e-2, i  # millimetres on |x| <= 1600
    valid = torch.tensor([1, 0] * (kp.shape[0] // 2), dtype=torch.int32).cuda()
    z = ops.convert_joints(kp, box, valid, None).cpu().numpy()
    assert np.abs(z[1::2]).max() == 0.0 and np.abs(z[0::2] - g["uvd_img"][0::2]).max() < 2e-4


def test_no_detection_branch_matches_reference_contract(fcos_sd, a2j_sd):
    """handnet_pipeline.py:107-108: nothing passes 0.7 -> (zeros[N,21,3] CPU, zeros_like(depth_images),
    zeros[N,4] float32 on the CPU)."""
    import types
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    sd = dict(fcos_sd)
    for k in ("head.classification_head.cls_logits.bias", "head.regression_head.bbox_ctrness.bias"):
        sd[k] = torch.full_like(sd[k], -20.0)        # every score far below 0.7
    net.detector.load_state_dict(sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    rgb, depth = synth.make_rgb(2, seed=1000), synth.make_depth(2, seed=2000).cuda()
    with torch.inference_mode():
        kp, db, crops = net([rgb[0].cuda(), rgb[1].cuda()], depth_images=depth)
    assert kp.device.type == "cpu" and kp.shape == (2, 21, 3) and float(kp.abs().max()) == 0.0
    assert db.shape == depth.shape and db.device == depth.device and float(db.abs().max()) == 0.0
    assert crops.device.type == "cpu" and crops.dtype == torch.float32 and crops.shape == (2, 4)
    # ros_demo.py:294 tests detections.max() == 0 for "no hand"
    assert crops.max() == 0


def test_handnet_follows_submodule_weight_reload(fcos_sd, a2j_sd):
    """net.detector.load_state_dict(...) AFTER a first forward must not leave HandNet on stale engines."""
    import types
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    rgb, depth = synth.make_rgb(1, seed=1000), synth.make_depth(1, seed=2000).cuda()
    with torch.inference_mode():
        kp0, _, _ = net([rgb[0].cuda()], depth_images=depth)
        e0 = net.engine()
        net.a2j.load_state_dict(synth.make_a2j_state_dict(seed=5), strict=False)
        kp1, _, _ = net([rgb[0].cuda()], depth_images=depth)
    assert net.engine() is not e0 and net.engine().a2j is net.a2j.engine()
    assert (kp1 - kp0).abs().max() > 1e-3


def test_many_frame_crop_box_parity(fcos_sd, a2j_sd):
    """Where end-to-end parity can actually break: 128 fresh frames (seeds 1000 / 2000 streams) through HIP and
    the oracle.  Integer crop boxes must be identical on EVERY frame (oracle.parity lists, for a failing frame, which
    coordinate or score sat at its decision boundary); keypoints stay < 1e-3."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    from oracle import parity
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    n = 128
    rgb, depth = synth.make_rgb(n, seed=1000), synth.make_depth(n, seed=2000)
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    stats, oracle_s, _ = parity.pipeline_parity(eng, rgb, depth, fcos_sd, a2j_sd, 3, chunk=8)
    print(f"[parity] {stats['crop_box_equal_frames']}/{n} equal crop boxes, max |dkp| {stats['max_abs_keypoint_diff']:.2e}, "
          f"oracle {oracle_s:.1f} s, flips: {stats['flips']}")
    assert stats["frames_with_hand"] == n
    assert stats["keypoints_within_tolerance"], stats["max_abs_keypoint_diff"]
    # fixed seeds: every one of the 128 integer crop boxes must be identical (a regression cannot hide in an allowance;
    # the per-flip diagnosis of oracle.parity is what the failure message shows)
    assert stats["crop_boxes_identical"] and stats["crop_box_equal_frames"] == n, stats["flips"]


def test_engine_range_check_raises_on_overflowing_activations(fcos_sd, a2j_sd):
    """HN_CHECK_RANGE=1 (HandNetEngine.check_range): a depth map in millimetres x 1000 drives A2J activations
    past 65504 -> RangeError instead of inf / NaN keypoints; ordinary inputs pass with the check on."""
    from hn_amd import ops, synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    eng.check_range = True
    rgb, depth = synth.make_rgb(2, seed=1000).cuda(), synth.make_depth(2, seed=2000).cuda()
    try:
        out = eng.forward_device(rgb, depth)
        assert torch.isfinite(out.keypoints).all()
        with pytest.raises(ops.RangeError):
            eng.forward_device(rgb, depth * 1.0e6)
        eng.check_range = False
        bad = eng.forward_device(rgb, depth * 1.0e6)          # what the check protects from: silently wrong numbers
        exact = A2JEngine(a2j_sd, device="cuda", precision="f32").forward_nhwc(bad.crops_nhwc, valid=bad.has_hand)
        rel = (bad.keypoints - exact).abs().max() / exact.abs().max()
        assert not torch.isfinite(bad.keypoints).all() or rel > 1e-2, rel
    finally:
        ops.range_check_enable(False)


def test_handnet_default_two_class_detector_matches_oracle(a2j_sd):
    """The constructor default is num_classes=2 (handnet_pipeline.py:47; hand class = 1): Cout = 4 head outputs,
    two-way argmax in the candidate kernel.  Crop boxes identical to the oracle, keypoints < 1e-3; also through the
    C++ layer graph (bit-identical to the Python engine)."""
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    from hn_amd.native_model import NativeModel
    from oracle import handnet_ref
    fsd = synth.make_fcos_state_dict(0, 2)
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"))          # num_classes=2 default
    assert net.num_classes == 2
    net.detector.load_state_dict(fsd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    rgb, depth = synth.make_rgb(3, seed=1000), synth.make_depth(3, seed=2000)
    with torch.inference_mode():
        kp, db, crops = net([r.cuda() for r in rgb], depth_images=depth.cuda())
    rkp, rdb, rcrops = handnet_ref.handnet_forward([r for r in rgb], depth, fsd, a2j_sd, 2)
    assert torch.equal(crops.cpu(), rcrops) and torch.equal(db.cpu(), rdb)
    assert (kp - rkp).abs().max().item() < 1e-3
    m = NativeModel(fsd, a2j_sd, num_classes=2)
    try:
        nkp, nbox, nhas = m.handnet(rgb.cuda(), depth.cuda())
        out = net.forward_device(rgb.cuda(), depth.cuda())
        assert torch.equal(nkp, out.keypoints) and torch.equal(nbox, out.crop_box) and torch.equal(nhas, out.has_hand)
    finally:
        m.close()


def test_handnet_refuses_nonfinite_keypoints(fcos_sd, a2j_sd):
    """Always-on safety net of the f16x3 path (no switch, no extra device work): the drop-in checks the keypoints it
    has just copied to the host and raises instead of returning inf / NaN.  (Non-finite values inside the networks are
    mostly laundered by the ReLUs -- max(NaN, 0) = 0 -- which is what HN_CHECK_RANGE=1 is for; what reaches the
    output is provoked here with a non-finite bias of the depth head's output convolution.)"""
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import ops, synth
    sd = {k: v.clone() for k, v in a2j_sd.items()}
    sd["DepthRegressionModel.output.bias"][5] = float("inf")
    args = types.SimpleNamespace(pretrained_fcos="unused.pth", pretrained_a2j="unused.pth")
    net = HandNet(args, num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    rgb, depth = synth.make_rgb(1, seed=1000).cuda(), synth.make_depth(1, seed=2000).cuda()
    with torch.inference_mode():
        assert not torch.isfinite(net.forward_device([rgb[0]], depth).keypoints).all()   # the engine itself returns them
        with pytest.raises(ops.RangeError, match="checkpoint"):
            net([rgb[0]], depth_images=depth)


def _dropin(fcos_sd, a2j_sd):
    from handnet_pipeline.handnet_pipeline import HandNet
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    return net.cuda().eval()


def test_nan_depth_pixels_give_nan_keypoints_like_the_reference(fcos_sd, a2j_sd):
    """ROS 32FC1 depth marks invalid pixels with NaN and the reference's caller passes them on (ros_demo.py:227-231): its
    network then returns NaN for every joint of a frame whose CROP holds one (oracle: the same plain torch ops) and finite
    keypoints for the other frames.  The drop-in must do the same -- not raise, not return finite garbage (the split
    convolutions' ReLUs map NaN to 0) -- eagerly, through the automatic graph replay, and through the C++ layer graph."""
    from hn_amd import synth
    from hn_amd.native_model import NativeModel
    from oracle import handnet_ref
    net = _dropin(fcos_sd, a2j_sd)
    rgb, depth = synth.make_rgb(3, seed=1000), synth.make_depth(3, seed=2000)
    clean = handnet_ref.handnet_forward(list(rgb), depth, fcos_sd, a2j_sd, 3)
    x1, y1, x2, y2 = clean[2][1].tolist()
    depth[1, 0, (y1 + y2) // 2, (x1 + x2) // 2] = float("nan")        # inside frame 1's crop
    bx = clean[2][2].tolist()
    oy, ox = (0, 0) if bx[1] > 8 and bx[0] > 8 else (479, 639)          # a pixel OUTSIDE frame 2's crop
    assert not (bx[0] <= ox <= bx[2] and bx[1] <= oy <= bx[3])
    depth[2, 0, oy, ox] = float("inf")
    rkp, rdb, rcrops = handnet_ref.handnet_forward(list(rgb), depth, fcos_sd, a2j_sd, 3)
    assert torch.isnan(rkp[1]).all() and torch.isfinite(rkp[0]).all() and torch.isfinite(rkp[2]).all()
    with torch.inference_mode():
        for call in range(6):                                           # eager, then captured + replayed
            kp, db, crops = net([r.cuda() for r in rgb], depth_images=depth.cuda())
            assert torch.isnan(kp[1]).all(), call
            assert (kp[[0, 2]] - rkp[[0, 2]]).abs().max().item() < 1e-3
            assert torch.equal(crops.cpu(), rcrops)
            assert torch.equal(torch.nan_to_num(db.cpu(), nan=-1.0), torch.nan_to_num(rdb, nan=-1.0))
    assert net.engine().graph_count() == 1
    out = net.forward_device(rgb.cuda(), depth.cuda(), _graph=False)
    assert out.has_hand.cpu().tolist() == [1, 2, 1]
    m = NativeModel(fcos_sd, a2j_sd, num_classes=3)
    try:
        nkp, nbox, nhas = m.handnet(rgb.cuda(), depth.cuda())
        assert nhas.cpu().tolist() == [1, 2, 1] and torch.isnan(nkp[1]).all()
        assert torch.equal(nkp[[0, 2]], out.keypoints[[0, 2]]) and torch.equal(nbox, out.crop_box)
    finally:
        m.close()


def test_nan_frame_does_not_switch_the_range_contract_off_for_the_other_frames(fcos_sd, a2j_sd, monkeypatch):
    """ADVICE r04 (low x2): one NaN depth pixel in one frame of a batch (normal for 32FC1 cameras) must not hide an overflow in
    another frame of the same call: frame 0 holds a NaN pixel inside its crop, frame 1's depth is finite but far outside the
    fp16 range -> ops.RangeError (it used to return silently).  And the sparse path reports has_hand == 2 like the dense one."""
    from hn_amd import ops, synth
    from oracle import handnet_ref
    net = _dropin(fcos_sd, a2j_sd)
    rgb, depth = synth.make_rgb(2, seed=1000), synth.make_depth(2, seed=2000)
    clean = handnet_ref.handnet_forward(list(rgb), depth, fcos_sd, a2j_sd, 3)
    x1, y1, x2, y2 = clean[2][0].tolist()
    depth[0, 0, (y1 + y2) // 2, (x1 + x2) // 2] = float("nan")
    with torch.inference_mode():
        kp, _db, _crops = net([r.cuda() for r in rgb], depth_images=depth.cuda())       # NaN alone: no error
        assert torch.isnan(kp[0]).all() and torch.isfinite(kp[1]).all()
        bad = depth.clone()
        bad[1] *= 1.0e6
        with pytest.raises(ops.RangeError, match="also saw non-finite"):
            net([r.cuda() for r in rgb], depth_images=bad.cuda())
    # sparse batch: 8 frames, a hand in 2 of them (the crop stage is patched as in test_sparse_stream_compacts_a2j), one of the two
    # with a NaN pixel in its crop: the compacted A2J call must report has_hand == 2 for it like the dense path does
    from hn_amd import pipeline
    eng = net.engine()
    rgb8, dep8 = synth.make_rgb(8, seed=1000).cuda(), synth.make_depth(8, seed=2000).cuda()
    bx = eng.forward_device(rgb8, dep8).crop_box[2].cpu().tolist()
    dep8[2, 0, (bx[1] + bx[3]) // 2, (bx[0] + bx[2]) // 2] = float("nan")
    keep = torch.zeros((8,), dtype=torch.int32, device="cuda")
    keep[[2, 5]] = 1
    real_crop = ops.crop_resize

    def sparse_crop(*a, **k):
        box, has, crops = real_crop(*a, **k)
        return box * keep[:, None].to(box.dtype), has * keep, crops * keep[:, None, None, None].to(crops.dtype)
    monkeypatch.setattr(pipeline.ops, "crop_resize", sparse_crop)
    dense = eng.forward_device(rgb8, dep8)
    assert dense.has_hand.cpu().tolist() == [0, 0, 2, 0, 0, 1, 0, 0]
    eng._sparse_hint, eng._hand_stat = True, None
    calls = []
    real_fwd = eng.a2j.forward_nhwc

    def spy(x, valid=None, **k):
        calls.append(x.shape[0])
        return real_fwd(x, valid=valid, **k)
    monkeypatch.setattr(eng.a2j, "forward_nhwc", spy)
    out = eng.forward_device(rgb8, dep8)
    assert calls == [2], calls
    assert out.has_hand.cpu().tolist() == [0, 0, 2, 0, 0, 1, 0, 0]
    assert torch.isnan(out.keypoints[2]).all() and torch.isfinite(out.keypoints[5]).all()
    assert (out.keypoints[5] - dense.keypoints[5]).abs().max().item() < 1e-4


def test_a2j_dropin_nan_crop_gives_nan_row(a2j_sd):
    """a2j.a2j.A2JModel (a2j_infer.py:25,59) on crops of which one holds a NaN pixel: NaN keypoints for that crop, like the
    reference's plain torch forward; the other crops are untouched."""
    from a2j.a2j import A2JModel
    from hn_amd import synth
    from oracle import a2j_ref
    model = A2JModel(21, crop_height=176, crop_width=176)
    model.load_state_dict(a2j_sd, strict=False)
    model = model.cuda().eval()
    x = synth.make_crops(3, seed=3000)
    x[2, 0, 17, 130] = float("nan")
    ref = a2j_ref.a2j_forward(x, a2j_sd)
    assert torch.isnan(ref[2]).all() and torch.isfinite(ref[:2]).all()
    with torch.inference_mode():
        kp = model(x.cuda())
    assert kp.device.type == "cpu" and torch.isnan(kp[2]).all()
    assert (kp[:2] - ref[:2]).abs().max().item() < 1e-3


def test_dropin_range_contract_is_always_on(fcos_sd, a2j_sd):
    """The f16x3 range contract needs no switch: HandNet.forward reads the step's flag words with the
    keypoints.  Depth in raw millimetres x 1000 (finite, far beyond 65504 after the first layers) raises ops.RangeError
    instead of returning finite-but-wrong keypoints (the ReLUs launder the NaNs: forward_device shows what would have
    been returned); saturated 16UC1 depth left unscaled (65535 > 65504) is refused as an INPUT; ordinary frames pass, also
    on the calls that replay a captured graph."""
    from hn_amd import _lib, ops, synth
    net = _dropin(fcos_sd, a2j_sd)
    assert net.engine().note_range
    rgb, depth = synth.make_rgb(1, seed=1000).cuda(), synth.make_depth(1, seed=2000).cuda()
    with torch.inference_mode():
        for _ in range(6):
            kp, _, _ = net([rgb[0]], depth_images=depth)
            assert torch.isfinite(kp).all()
        out = net.forward_device(rgb, depth, _graph=False)
        assert out.range_flags.cpu().tolist() == [0, 0, 0, 0]
        # a depth scale that keeps the INPUT in range (max 1.5 x scale < 65504) but not the activations behind it
        scale = None
        for sc in (1.0e4, 2.0e4, 4.0e4):
            words = net.forward_device(rgb, depth * sc, _graph=False).range_flags.cpu().tolist()
            if ops.range_bits(words) == _lib.RANGE_ACTIVATION:
                scale = sc
                break
        assert scale is not None, words
        with pytest.raises(ops.RangeError, match="activation"):
            net([rgb[0]], depth_images=depth * scale)
        with pytest.raises(ops.RangeError, match="METRES"):
            net([rgb[0]], depth_images=torch.full_like(depth, 65535.0))
        kp, _, _ = net([rgb[0]], depth_images=depth)                      # the flags are per step: the next call is clean
        assert torch.isfinite(kp).all()


def test_sparse_stream_compacts_a2j(fcos_sd, a2j_sd, monkeypatch):
    """Sparse streams: once a step has had a hand in fewer than half of its frames, the next eager
    step runs A2J on the frames with a hand only -- same keypoints on those frames, zero rows elsewhere; the dense path
    never synchronises for it, and graph capture never takes it."""
    from hn_amd import ops, pipeline, synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    eng = pipeline.HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    n = 16
    rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
    dense = eng.forward_device(rgb, depth)
    torch.cuda.synchronize()
    assert int(dense.has_hand.sum()) == n and not eng._sparse_hint
    keep = torch.zeros((n,), dtype=torch.int32, device="cuda")
    keep[[1, 6, 11]] = 1
    real_crop = ops.crop_resize
    calls = []

    def sparse_crop(*a, **k):          # frames 1, 6, 11 keep their hand, the others lose it (as an empty scene would)
        box, has, crops = real_crop(*a, **k)
        has = has * keep
        return box * keep[:, None].to(box.dtype), has, crops * keep[:, None, None, None].to(crops.dtype)
    monkeypatch.setattr(pipeline.ops, "crop_resize", sparse_crop)
    real_fwd = eng.a2j.forward_nhwc

    def spy(x, valid=None, **k):
        calls.append(x.shape[0])
        return real_fwd(x, valid=valid, **k)
    monkeypatch.setattr(eng.a2j, "forward_nhwc", spy)
    first = eng.forward_device(rgb, depth)        # still the masked full-batch form; its count arms the hint
    torch.cuda.synchronize()
    second = eng.forward_device(rgb, depth)       # compacted
    assert calls == [n, 3], calls
    assert torch.equal(second.has_hand, first.has_hand) and int(second.has_hand.sum()) == 3
    assert torch.equal(second.keypoints[keep == 0], torch.zeros_like(second.keypoints[keep == 0]))
    sel = keep.bool()
    assert (second.keypoints[sel] - first.keypoints[sel]).abs().max().item() < 1e-4
    assert (second.keypoints[sel] - dense.keypoints[sel]).abs().max().item() < 1e-4


def test_dropin_forward_switches_itself_to_graph_replay(fcos_sd, a2j_sd):
    """handnet_pipeline.HandNet.forward as ros_demo.py:270-273 calls it, one frame per call: after AUTO_GRAPH_CALLS same-shape
    calls it captures the step and replays it -- same kernels, so the results stay bit-identical to the eager calls, they are
    fresh tensors (not views of the captured buffers), a new input shape falls back to eager, and enable_graph(False) keeps
    the whole thing off."""
    import types
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    frames = [(synth.make_rgb(1, seed=1000 + i).cuda(), synth.make_depth(1, seed=2000 + i).cuda()) for i in range(3)]
    with torch.inference_mode():
        eager = [net([rgb[0]], depth_images=dep) for rgb, dep in frames]            # calls 1-3: eager
        assert net.engine().graph_count() == 0
        held = net([frames[0][0][0]], depth_images=frames[0][1])                     # call 4: still eager (streak == 3)
        again = [net([rgb[0]], depth_images=dep) for rgb, dep in frames]            # calls 5-7: captured, then replayed
        assert net.engine().graph_count() == 1
        for (kp, db, box), (kp2, db2, box2) in zip(eager, again):
            assert kp2.device.type == "cpu" and torch.equal(kp, kp2) and torch.equal(db, db2) and torch.equal(box, box2)
        assert torch.equal(held[0], eager[0][0]) and torch.equal(held[1], eager[0][1])   # earlier results are not overwritten
        first = again[0][1].clone()
        net([frames[1][0][0]], depth_images=frames[1][1])
        assert torch.equal(again[0][1], first)                                         # ... nor are results of replayed calls
        wide = synth.make_rgb(1, seed=5).cuda()[:, :, :400, :]
        kp_w, _, _ = net([wide[0]], depth_images=frames[0][1][:, :, :400, :].contiguous())   # another shape: eager again
        assert net.engine().graph_count() == 1 and tuple(kp_w.shape) == (1, 21, 3)
        net.enable_graph(False)
        for rgb, dep in frames * 2:
            net([rgb[0]], depth_images=dep)
        assert net.engine().graph_count() == 1


def test_pipeline_is_deterministic_over_many_steps(fcos_sd, a2j_sd):
    """A race in any of the hand-synchronised kernels (LDS rings, counted waits, chunked compaction, fused GroupNorm apply)
    would show as run-to-run differences: 60 steps of the full batch-32 pipeline and 200 of the single-frame pipeline on
    fixed inputs, every output bit-identical to the first step's."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    for n, steps in ((32, 60), (1, 200)):
        rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
        ref = eng.forward_device(rgb, depth)
        kp0, box0, has0 = ref.keypoints.clone(), ref.crop_box.clone(), ref.has_hand.clone()
        bad = torch.zeros((), device="cuda", dtype=torch.int64)
        for _ in range(steps):
            out = eng.forward_device(rgb, depth)
            bad += (out.keypoints != kp0).sum() + (out.crop_box != box0).sum() + (out.has_hand != has0).sum()
        assert int(bad) == 0


# ---- raw camera frames: the reference caller's ingest (ros_demo.py:227-231,266-269) as one kernel ----
def _raw_frames(n, h=480, w=640, seed=7):
    rng = np.random.default_rng(seed)
    bgr = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    mm = rng.integers(300, 1500, size=(n, h, w)).astype(np.uint16)
    return bgr, mm


def _host_ingest(bgr, mm):
    """What ros_demo.py:230-231,266 compute on the host (cv2.COLOR_BGR2RGB = channel reversal)."""
    rgb = np.ascontiguousarray(bgr[..., ::-1]).transpose(0, 3, 1, 2).astype(np.float32) / 255.0
    depth = mm.astype(np.float32)
    depth /= 1000.0
    return torch.from_numpy(np.ascontiguousarray(rgb)), torch.from_numpy(depth).unsqueeze(1)


@pytest.mark.parametrize("shape", [(2, 480, 640), (1, 37, 53), (3, 30, 44)])
def test_ingest_raw_is_bit_identical_to_the_host_arithmetic(shape):
    """hn_ingest_u8bgr_u16mm from device memory and straight from PINNED host memory: uint8 / 255.0 and uint16 / 1000.0 are IEEE
    divisions of exact integers, so the fp32 tensors must equal numpy's bit for bit -- the vector path (H*W % 4 == 0) and
    the per-pixel path; 32FC1 depth passes through; the RGB-D tensor is cat([rgb, depth])."""
    from hn_amd import ops
    n, h, w = shape
    bgr, mm = _raw_frames(n, h, w)
    ref_rgb, ref_d = _host_ingest(bgr, mm)
    tb, td = torch.from_numpy(bgr), torch.from_numpy(mm)
    for src_b, src_d in ((tb.cuda(), td.cuda()), (tb.pin_memory(), td.pin_memory())):
        rgb, d, rgbd = ops.ingest_raw(src_b, src_d, want_rgbd=True)
        assert torch.equal(rgb.cpu(), ref_rgb) and torch.equal(d.cpu(), ref_d)
        assert torch.equal(rgbd.cpu(), torch.cat([ref_rgb, ref_d], dim=1))
    f32 = ref_d[:, 0].contiguous()
    f32[0, 0, 0] = float("nan")                                   # 32FC1 marks invalid pixels so: passed through untouched
    rgb, d, _ = ops.ingest_raw(tb.cuda(), f32.cuda())
    assert torch.equal(torch.nan_to_num(d.cpu(), nan=-1.0), torch.nan_to_num(f32.unsqueeze(1), nan=-1.0))
    rgb_only, none_d, _ = ops.ingest_raw(tb.cuda())
    assert none_d is None and torch.equal(rgb_only.cpu(), ref_rgb)
    with pytest.raises(RuntimeError, match="pinned"):
        ops.ingest_raw(tb, td.cuda())
    with pytest.raises(TypeError):
        ops.ingest_raw(tb.cuda().float(), td.cuda())


def test_forward_raw_equals_forward_on_the_host_converted_frames(fcos_sd, a2j_sd):
    """HandNet.forward_raw(bgr8 frames, 16UC1 depth) == HandNet.forward(the frames converted on the host exactly as
    ros_demo.py:230-231,266 converts them): same kernels behind one ingest launch, so every returned tensor is bit-identical --
    from numpy arrays (pageable host memory), pinned tensors and device tensors, eagerly and through the captured step."""
    net = _dropin(fcos_sd, a2j_sd)
    ref_net = _dropin(fcos_sd, a2j_sd).enable_graph(False)
    frames = [_raw_frames(2, seed=11 + i) for i in range(3)]
    with torch.inference_mode():
        for call in range(7):                                   # calls 1-4 eager, then captured + replayed
            bgr, mm = frames[call % 3]
            rgb, depth = _host_ingest(bgr, mm)
            want = ref_net([r.cuda() for r in rgb], depth_images=depth.cuda())
            if call % 3 == 0:
                got = net.forward_raw(bgr, mm)                                          # numpy, pageable
            elif call % 3 == 1:
                got = net.forward_raw(torch.from_numpy(bgr).pin_memory(), torch.from_numpy(mm).pin_memory())
            else:
                got = net.forward_raw(torch.from_numpy(bgr).cuda(), torch.from_numpy(mm).cuda())
            assert got[0].device.type == "cpu" and got[1].is_cuda and got[2].dtype == torch.int64
            for a, b in zip(got, want):
                assert torch.equal(a.cpu(), b.cpu()), call
        assert net.engine().graph_count() == 1
        one = net.forward_raw(frames[0][0][0], frames[0][1][0])                      # a single [H,W,3] frame + [H,W] depth
        assert tuple(one[0].shape) == (1, 21, 3)
        # nothing detected: the placeholder tuple of handnet_pipeline.py:107-108 with the fp32 depth shape
        black = net.forward_raw(np.zeros((2, 480, 640, 3), np.uint8), frames[0][1])
        if float(black[2].abs().max()) == 0.0:
            assert tuple(black[0].shape) == (2, 21, 3) and tuple(black[1].shape) == (2, 1, 480, 640) and tuple(black[2].shape) == (2, 4)


def test_auto_captured_graphs_are_evicted_least_recently_used(fcos_sd, a2j_sd):
    """a caller that sweeps input shapes keeps at most AUTO_GRAPH_MAX_SHAPES captured steps (each owns a
    static activation pool); one more shape evicts the least recently used capture instead of refusing to capture."""
    from hn_amd import synth
    net = _dropin(fcos_sd, a2j_sd)
    net.AUTO_GRAPH_MAX_SHAPES = 2
    net.AUTO_GRAPH_CALLS = 1
    eng = net.engine()

    def run(h, w, times=3):
        rgb, depth = synth.make_rgb(1, seed=1000)[:, :, :h, :w].contiguous().cuda(), synth.make_depth(1, seed=2000)[:, :, :h, :w].contiguous().cuda()
        with torch.inference_mode():
            for _ in range(times):
                out = net([rgb[0]], depth_images=depth)
        return out
    a = run(480, 640)
    run(448, 640)
    assert eng.graph_count() == 2 and eng.has_graph((1, 3, 480, 640), (1, 1, 480, 640), to_host=True)
    run(480, 640, times=1)                                     # a use: 448 x 640 is now the least recently used
    run(416, 640)
    assert eng.graph_count() == 2
    assert eng.has_graph((1, 3, 480, 640), (1, 1, 480, 640), to_host=True) and not eng.has_graph((1, 3, 448, 640), (1, 1, 448, 640), to_host=True)
    b = run(480, 640, times=1)
    for x, y in zip(a, b):
        assert torch.equal(x.cpu(), y.cpu())


@pytest.mark.parametrize("n", [1, 6])
def test_round5_kernel_forms_are_bit_identical_end_to_end(fcos_sd, a2j_sd, n):
    """The three kernel routes added in round 5 -- the streaming 1x1 kernel (FPN P3 lateral from batch 5 on), the mixed-tile grouped
    launch (tower layers at batch 1-2) the deep-k form of the 64x64 tile (ResNet-34 layer3 at batch 1) and the split-K reduction inside the last workgroup -- keep every k order:
    the whole pipeline's outputs (detections, crop boxes, keypoints) are bit-identical with the routes switched off by name."""
    from hn_amd import ops, synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    ops.clear_plan_caches()
    new = eng.forward_device(rgb, depth)
    forms = ("conv_no_stream", "conv_no_mixed", "conv_no_deepk", "conv_no_fused_reduce")
    for f in forms:
        ops.set_form(f, True)
    try:
        ops.clear_plan_caches()
        old = eng.forward_device(rgb, depth)
    finally:
        for f in forms:
            ops.set_form(f, False)
        ops.clear_plan_caches()
    assert torch.equal(new.keypoints.view(torch.int32), old.keypoints.view(torch.int32))
    assert torch.equal(new.crop_box, old.crop_box) and torch.equal(new.has_hand, old.has_hand)
    assert torch.equal(new.detections.count, old.detections.count)
    for i, k in enumerate(new.detections.count.cpu().tolist()):      # (rows at or beyond count[i] are undefined)
        assert torch.equal(new.detections.boxes[i, :k], old.detections.boxes[i, :k])
        assert torch.equal(new.detections.scores[i, :k], old.detections.scores[i, :k])


def test_engine_forward_raw_rgbd_equals_forward_device(fcos_sd, a2j_rgbd_sd):
    """The RGB-D model through the raw-frame entry: the ingest kernel writes the 4-channel tensor `cat([rgb, depth], 1)` that
    ros_demo.py:268-270 builds on the host; keypoints, crop boxes and the 4-channel crops (channel order [2,1,0,3],
    handnet_pipeline.py:102) equal forward_device on the host-converted inputs bit for bit, eagerly and through a captured step."""
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_rgbd_sd, rgbd=True, device="cuda"), 3)
    bgr, mm = _raw_frames(2, seed=21)
    rgb, depth = _host_ingest(bgr, mm)
    want = eng.forward_device(rgb.cuda(), torch.cat([rgb, depth], dim=1).cuda())
    tb, td = torch.from_numpy(bgr), torch.from_numpy(mm)
    for use_graph in (False, True, True):
        got = eng.forward_raw(tb, td, use_graph=use_graph)
        torch.cuda.synchronize()
        assert torch.equal(got.keypoints.view(torch.int32), want.keypoints.view(torch.int32))
        assert torch.equal(got.crop_box, want.crop_box) and torch.equal(got.crops_nhwc, want.crops_nhwc)
    assert eng.graph_count() == 1


def test_exact_f32_mode_propagates_nan_crops_like_torch(a2j_sd):
    """precision="f32" has no stem image to mark a crop on: a NaN pixel travels through the network itself.  With NaN-propagating
    ReLU and max pooling (round 5; torch.relu / max_pool2d keep NaN, v_max_f32 would not) every joint of that crop comes out NaN,
    as the reference's plain torch forward returns it, and the other crops are untouched."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from oracle import a2j_ref
    eng = A2JEngine(a2j_sd, device="cuda", precision="f32")
    x = synth.make_crops(3, seed=3200)
    clean = eng.forward(x.cuda()).cpu()
    x[1, 0, 90, 40] = float("nan")
    ref = a2j_ref.a2j_forward(x, a2j_sd)
    got = eng.forward(x.cuda()).cpu()
    assert torch.isnan(ref[1]).all() and torch.isnan(got[1]).all()
    assert torch.equal(got[[0, 2]], clean[[0, 2]]) and (got[[0, 2]] - ref[[0, 2]]).abs().max().item() < 1e-3


def test_nonfinite_rgb_pixels_raise_instead_of_giving_undefined_detections(fcos_sd, a2j_sd):
    """A NaN RGB pixel (not a camera's case: frames arrive as uint8) would make the reference's detector return nothing for that
    frame; the split-precision detector cannot promise that, so the drop-in is LOUD: the NaN travels through the NaN-propagating
    ReLUs into the activations, the range contract flags it, HandNet.forward raises ops.RangeError naming the cause (DESIGN section
    2; a documented deviation) -- never silently undefined detections."""
    from hn_amd import ops, synth
    net = _dropin(fcos_sd, a2j_sd)
    rgb, depth = synth.make_rgb(2, seed=1000), synth.make_depth(2, seed=2000)
    rgb[1, 0, 200, 300] = float("nan")
    with torch.inference_mode():
        with pytest.raises(ops.RangeError, match="non-finite RGB"):
            net([r.cuda() for r in rgb], depth_images=depth.cuda())
        rgb[1, 0, 200, 300] = 0.5
        kp, _, _ = net([r.cuda() for r in rgb], depth_images=depth.cuda())       # the flags are per step
        assert torch.isfinite(kp).all()


def test_engine_forward_raw_back_to_back_from_pageable_memory(fcos_sd, a2j_sd):
    """HandNetEngine.forward_raw is sync-free, and pageable inputs travel through a pinned staging buffer that the
    ingest kernel reads ASYNCHRONOUSLY.  Five calls in a row with five different frame pairs and NO synchronisation in between
    (a pipelined caller): every step's results must be those of its own frames -- the staging buffers rotate and a buffer is
    only rewritten once the ingest launch that read it has passed its event."""
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    frames = [_raw_frames(2, seed=31 + i) for i in range(5)]
    refs = []
    for bgr, mm in frames:
        rgb, depth = _host_ingest(bgr, mm)
        o = eng.forward_device(rgb.cuda(), depth.cuda())
        torch.cuda.synchronize()
        refs.append((o.keypoints.clone(), o.crop_box.clone()))
    outs = [eng.forward_raw(torch.from_numpy(bgr), torch.from_numpy(mm)) for bgr, mm in frames]     # no sync in between
    torch.cuda.synchronize()
    for o, (kp, box) in zip(outs, refs):
        assert torch.equal(o.crop_box, box) and torch.equal(o.keypoints, kp)
    assert len({tuple(r[1].flatten().tolist()) for r in refs}) > 1          # (the frames really differ in their results)
    ring = next(iter(eng._raw_staging.values()))
    assert len(ring["slots"]) == 2

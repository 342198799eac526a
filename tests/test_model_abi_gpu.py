"""Model-level C ABI (hn_create / hn_load_weight / hn_finalize / hn_*_forward / hn_destroy): the C++ layer graphs
must reproduce the Python engines bit for bit (same launches, same descriptors), and fail loudly on bad use."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native(fcos_sd, a2j_sd):
    from hn_amd.native_model import NativeModel
    m = NativeModel(fcos_sd, a2j_sd, num_classes=3)
    yield m
    m.close()


def test_handnet_forward_equals_python_engine(native, fcos_sd, a2j_sd):
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    for n, seed in ((2, 1000), (5, 1100), (2, 1000)):          # a larger batch re-sizes the arena; then a plan hit
        rgb, depth = synth.make_rgb(n, seed=seed).cuda(), synth.make_depth(n, seed=seed + 1000).cuda()
        ref = eng.forward_device(rgb, depth)
        kp, box, has = native.handnet(rgb, depth)
        assert torch.equal(box, ref.crop_box) and torch.equal(has, ref.has_hand)
        assert torch.equal(kp, ref.keypoints)                    # bit-identical: the same kernels in the same order


def test_rgbd_handnet_forward_equals_python_engine(fcos_sd, a2j_rgbd_sd):
    """RGB-D variant through the C++ graph: 4-channel A2J stem, 4-channel crops with the [2,1,0,3] permutation
    (handnet_pipeline.py:102), depth_images = cat([rgb, depth]) as ros_demo.py:268-270 passes them."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.native_model import NativeModel
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_rgbd_sd, rgbd=True, device="cuda"), 3)
    m = NativeModel(fcos_sd, a2j_rgbd_sd, num_classes=3, rgbd=True)
    try:
        rgb, depth = synth.make_rgb(3, seed=1000).cuda(), synth.make_depth(3, seed=2000).cuda()
        rgbd = torch.cat([rgb, depth], 1).contiguous()
        ref = eng.forward_device(rgb, rgbd)
        kp, box, has = m.handnet(rgb, rgbd)
        assert torch.equal(box, ref.crop_box) and torch.equal(has, ref.has_hand) and torch.equal(kp, ref.keypoints)
        with pytest.raises(RuntimeError, match="hn_handnet_forward"):
            m.a2j(depth[:, :, :176, :176].contiguous())      # the A2J-only entry takes 1-channel crops
    finally:
        m.close()


def test_fcos_forward_equals_python_engine(native, fcos_sd):
    from hn_amd import synth
    from hn_amd.fcos_engine import FCOSEngine
    eng = FCOSEngine(fcos_sd, 3, device="cuda")
    rgb = synth.make_rgb(2, seed=1000).cuda()
    det, _ = eng.detect(rgb)
    boxes, scores, labels, sides, level, count = native.fcos(rgb)
    assert native.fcos_capacity(480, 640) == 17850 == det.scores.shape[1]
    assert torch.equal(count, det.count)
    for i, k in enumerate(count.tolist()):
        assert k > 0
        assert torch.equal(boxes[i, :k], det.boxes[i, :k]) and torch.equal(scores[i, :k], det.scores[i, :k])
        assert torch.equal(labels[i, :k], det.labels[i, :k]) and torch.equal(sides[i, :k], det.sides[i, :k])
        assert torch.equal(level[i, :k], det.level[i, :k])


def test_fcos_ext_forward_equals_python_engine():
    """ext=True (the FCOS class default, trainval_net_fcos.py --test-only): contact state and dxdy magnitudes of the
    kept detections through the C++ graph == the Python engine."""
    from hn_amd import synth
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.native_model import NativeModel
    sd = synth.make_fcos_state_dict(0, 3, ext=True)
    eng = FCOSEngine(sd, 3, device="cuda", ext=True)
    rgb = synth.make_rgb(2, seed=1000).cuda()
    det, _, contacts, dxdymags = eng.detect_ext(rgb)
    m = NativeModel(sd, None, num_classes=3, ext=True)
    try:
        boxes, scores, labels, sides, level, count, ncon, ndx = m.fcos_ext(rgb)
        assert torch.equal(count, det.count)
        for i, k in enumerate(count.tolist()):
            assert k > 0 and torch.equal(boxes[i, :k], det.boxes[i, :k]) and torch.equal(labels[i, :k], det.labels[i, :k])
            assert torch.equal(ncon[i, :k], contacts[i, :k]) and torch.equal(ndx[i, :k], dxdymags[i, :k])
        plain = NativeModel(synth.make_fcos_state_dict(0, 3), None, num_classes=3)
        try:
            with pytest.raises(RuntimeError, match="ext = 1"):
                plain.fcos_ext(rgb)
        finally:
            plain.close()
    finally:
        m.close()


def test_a2j_forward_equals_python_engine_and_golden(native, a2j_sd, golden_dir):
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    eng = A2JEngine(a2j_sd, device="cuda")
    x = synth.make_crops(3, 176, seed=3000).cuda()
    assert torch.equal(native.a2j(x), eng.forward(x))
    valid = torch.tensor([1, 0, 1], dtype=torch.int32, device="cuda")
    kp = native.a2j(x, valid)
    assert float(kp[1].abs().max()) == 0.0 and torch.equal(kp[0], eng.forward(x)[0])
    g = np.load(golden_dir / "a2j_forward.npz")                  # the reference's own output on the same seeded crops
    x2 = synth.make_crops(2, 176, seed=int(g["input_seed"])).cuda()
    assert np.abs(native.a2j(x2).cpu().numpy() - g["keypoints"]).max() < 1e-3


def test_a2j_forward_nan_crop_gives_nan_row_like_the_python_engine(native, a2j_sd):
    """hn_a2j_forward keeps flags of its own, so a crop with a NaN / inf pixel returns a NaN row (what
    a2j/a2j.py:243-250 returns for it) instead of finite numbers from laundered NaNs -- with and without the caller's `valid`,
    on the multi-launch path (<= 4 crops) and the grouped one; rows of the other crops bit-identical to A2JEngine.forward."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    eng = A2JEngine(a2j_sd, device="cuda")
    for k in (3, 6):
        x = synth.make_crops(k, 176, seed=3100 + k)
        x[1, 0, 40, 99] = float("nan")
        x[k - 1, 0, 0, 0] = float("inf")
        x = x.cuda()
        ref = eng.forward(x)
        assert torch.isnan(ref[1]).all() and torch.isnan(ref[k - 1]).all() and torch.isfinite(ref[0]).all()
        kp = native.a2j(x)
        assert torch.equal(torch.nan_to_num(kp, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
        valid = torch.ones((k,), device="cuda", dtype=torch.int32)
        valid[0] = 0
        kpv = native.a2j(x, valid)
        assert valid.cpu().tolist() == [0] + [1] * (k - 1)                  # the caller's flags stay read-only
        assert (kpv[0] == 0).all() and torch.isnan(kpv[1]).all() and torch.isnan(kpv[k - 1]).all()
        if k > 3:
            assert torch.equal(kpv[2], ref[2])


def test_model_abi_errors_are_loud(a2j_sd):
    from hn_amd import _lib
    from hn_amd.native_model import NativeModel
    lib = _lib.load()
    sd = {k: v for k, v in a2j_sd.items() if k != "Backbone.model.layer2.0.bn1.running_var"}
    with pytest.raises(RuntimeError, match="layer2.0.bn1.running_var"):
        NativeModel(None, sd)
    m = NativeModel(None, a2j_sd)
    try:
        with pytest.raises(RuntimeError, match="HN_MODEL_FCOS"):
            m.fcos(torch.zeros((1, 3, 480, 640), device="cuda"))
        t = torch.zeros(4)
        st = lib.hn_load_weight(m._h, b"x", t.data_ptr(), (C.c_int64 * 1)(4), 1)
        assert st != 0 and b"after hn_finalize" in lib.hn_last_error()
    finally:
        m.close()
    big = dict(a2j_sd)
    big["regressionModel.conv2.weight"] = a2j_sd["regressionModel.conv2.weight"] * 1e6   # leaves the fp16 range once folded
    with pytest.raises(RuntimeError, match="fp16 range"):
        NativeModel(None, big)


def test_f16x1_mode_through_the_model_abi(native, fcos_sd, a2j_sd):
    """hn_model_config.f16_terms = 1: the throughput mode (hi*hi term only) through the C++ layer graphs -- the same
    launches as the Python engines built with precision="f16x1" (bit-identical), and measurably NOT the default mode."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.native_model import NativeModel
    from hn_amd.pipeline import HandNetEngine
    eng = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda", precision="f16x1"),
                        A2JEngine(a2j_sd, device="cuda", precision="f16x1"), 3)
    m = NativeModel(fcos_sd, a2j_sd, num_classes=3, precision="f16x1")
    try:
        rgb, depth = synth.make_rgb(3, seed=1000).cuda(), synth.make_depth(3, seed=2000).cuda()
        ref = eng.forward_device(rgb, depth)
        kp, box, has = m.handnet(rgb, depth)
        assert torch.equal(box, ref.crop_box) and torch.equal(has, ref.has_hand) and torch.equal(kp, ref.keypoints)
        kp3, box3, has3 = native.handnet(rgb, depth)
        assert torch.equal(has3, has)
        same = (box3 == box).all(dim=1)
        d = (kp3 - kp)[same].abs().max().item()
        assert 1e-4 < d < 1.0, d                       # fp16-grade, not fp32-grade: ~0.01-0.1 px on equal crop boxes
    finally:
        m.close()
    with pytest.raises(RuntimeError, match="f16_terms"):
        from hn_amd import _lib
        cfg = _lib.ModelConfig(parts=_lib.MODEL_A2J, num_classes=3, num_joints=21, f16_terms=2)
        h = C.c_void_p()
        _lib.check(_lib.load().hn_create(C.byref(cfg), C.byref(h)), "hn_create")


def test_fcos_forward_list_equals_python_engine(native, fcos_sd):
    """hn_fcos_forward_list: differently sized images through the C++ layer graph (torchvision batch_images semantics,
    fcos.py:702-709: each image resized on its own into the common canvas, boxes rescaled per image) == FCOSEngine.detect
    on the same list, bit for bit; and a list of equal sizes == the batched entry point."""
    from hn_amd import synth
    from hn_amd.fcos_engine import FCOSEngine
    eng = FCOSEngine(fcos_sd, 3, device="cuda")
    full = synth.make_rgb(3, seed=1200).cuda()
    images = [full[0], full[1][:, :400, :560].contiguous(), full[2][:, :300, :].contiguous()]
    det, _ = eng.detect(images)
    boxes, scores, labels, sides, level, count = native.fcos_list(images)
    assert boxes.shape[1] == det.scores.shape[1] and torch.equal(count, det.count)
    for i, k in enumerate(count.tolist()):
        assert torch.equal(boxes[i, :k], det.boxes[i, :k]) and torch.equal(scores[i, :k], det.scores[i, :k])
        assert torch.equal(labels[i, :k], det.labels[i, :k]) and torch.equal(level[i, :k], det.level[i, :k])
    assert int(count.max()) > 0
    same = [full[0], full[1], full[2]]
    a, b = native.fcos_list(same), native.fcos(full)
    assert torch.equal(a[5], b[5])
    for i, k in enumerate(a[5].tolist()):
        assert torch.equal(a[0][i, :k], b[0][i, :k]) and torch.equal(a[1][i, :k], b[1][i, :k])
    with pytest.raises(RuntimeError, match="hn_fcos_forward_list"):
        i32 = dict(device="cuda", dtype=torch.int32)
        hs, ws, ptrs = (C.c_int32 * 1)(480), (C.c_int32 * 1)(640), (C.c_void_p * 1)(full[0].data_ptr())
        bad = torch.zeros((1, 7, 4), device="cuda")
        from hn_amd._lib import check
        check(native.lib.hn_fcos_forward_list(native._h, ptrs, hs, ws, 1, bad.data_ptr(), bad.data_ptr(), bad.data_ptr(),
                                              bad.data_ptr(), bad.data_ptr(), bad.data_ptr(), 7, None), "hn_fcos_forward_list")


def test_f32_mode_equals_python_engines_and_the_oracle(fcos_sd, a2j_sd):
    """hn_model_config.precision = HN_PRECISION_F32 (VERDICT r03 missing #4: the reference's own arithmetic for a C host):
    the C++ graphs issue the launches of FCOSEngine / A2JEngine(precision="f32") -- bit-identical tensors for the detector,
    the A2J-only entry, the whole pipeline and a mixed-size image list -- and the mode agrees with the oracle like the
    default one does (crop boxes identical, keypoints < 1e-3)."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.native_model import NativeModel
    from hn_amd.pipeline import HandNetEngine
    from oracle import handnet_ref
    fe, ae = FCOSEngine(fcos_sd, 3, device="cuda", precision="f32"), A2JEngine(a2j_sd, device="cuda", precision="f32")
    eng = HandNetEngine(fe, ae, 3)
    m = NativeModel(fcos_sd, a2j_sd, num_classes=3, precision="f32")
    try:
        rgb, depth = synth.make_rgb(2, seed=1000).cuda(), synth.make_depth(2, seed=2000).cuda()
        ref = eng.forward_device(rgb, depth)
        kp, box, has = m.handnet(rgb, depth)
        assert torch.equal(box, ref.crop_box) and torch.equal(has, ref.has_hand) and torch.equal(kp, ref.keypoints)
        rkp, _, rcrops = handnet_ref.handnet_forward([r.cpu() for r in rgb], depth.cpu(), fcos_sd, a2j_sd, 3)
        assert torch.equal(box.cpu(), rcrops) and (kp.cpu() - rkp).abs().max().item() < 1e-3
        det, _ = fe.detect(rgb)
        boxes, scores, labels, sides, level, count = m.fcos(rgb)
        assert torch.equal(count, det.count)
        for i, k in enumerate(count.tolist()):
            assert k > 0 and torch.equal(boxes[i, :k], det.boxes[i, :k]) and torch.equal(scores[i, :k], det.scores[i, :k])
            assert torch.equal(labels[i, :k], det.labels[i, :k])
        crops = synth.make_crops(3, seed=3000).cuda()
        assert torch.equal(m.a2j(crops), ae.forward(crops))
        imgs = [synth.make_rgb(1, 480, 640, seed=7)[0].cuda(), synth.make_rgb(1, 360, 600, seed=8)[0].cuda()]
        det_l, _ = fe.detect(imgs)
        lb, ls_, ll, _, _, lc = m.fcos_list(imgs)
        assert torch.equal(lc, det_l.count)
        for i, k in enumerate(lc.tolist()):
            assert torch.equal(lb[i, :k], det_l.boxes[i, :k]) and torch.equal(ll[i, :k], det_l.labels[i, :k])
    finally:
        m.close()


def test_model_config_image_mean_std(fcos_sd):
    """hn_model_config.image_mean / image_std (the FCOS ctor's, fcos.py:501-505) reach the C++ graph's transform: same
    detections as FCOSEngine with the same normalisation, in both precision modes; half-given std is refused."""
    from hn_amd import synth
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.native_model import NativeModel
    mean, std = [0.40, 0.50, 0.45], [0.20, 0.25, 0.30]
    rgb = synth.make_rgb(2, seed=1234).cuda()
    for precision in ("f16x3", "f32"):
        eng = FCOSEngine(fcos_sd, 3, device="cuda", precision=precision, image_mean=mean, image_std=std)
        det, _ = eng.detect(rgb)
        m = NativeModel(fcos_sd, None, num_classes=3, precision=precision, image_mean=mean, image_std=std)
        try:
            boxes, scores, labels, _, _, count = m.fcos(rgb)
            assert torch.equal(count, det.count)
            for i, k in enumerate(count.tolist()):
                assert torch.equal(boxes[i, :k], det.boxes[i, :k]) and torch.equal(labels[i, :k], det.labels[i, :k])
        finally:
            m.close()
    with pytest.raises(RuntimeError, match="image_std"):
        NativeModel(fcos_sd, None, num_classes=3, image_mean=mean, image_std=[0.2, 0.0, 0.3])


def test_handnet_forward_xyz_carries_the_converted_joints(native):
    """hn_handnet_forward_xyz: the same launches as hn_handnet_forward (keypoints / boxes / flags identical), plus image (u,v,d)
    and camera xyz written by the aggregation's epilogue -- bit-identical to hn_convert_joints_f32 on its results, with and
    without the live caller's clamps; a batch above the multi-launch threshold too (another A2J graph)."""
    from hn_amd import ops, synth
    paras = (617.343, 617.343, 312.42, 241.42)
    for n in (2, 6):
        rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
        kp0, box0, has0 = native.handnet(rgb, depth)
        kp, img, xyz, box, has = native.handnet_xyz(rgb, depth, paras)
        assert torch.equal(kp, kp0) and torch.equal(box, box0) and torch.equal(has, has0) and int((has == 1).sum()) == n
        assert torch.equal(img, ops.convert_joints(kp, box, has, None)) and torch.equal(xyz, ops.convert_joints(kp, box, has, paras))
        kp2, img2, xyz2, _b, _h = native.handnet_xyz(rgb, depth, None, clamp=True)
        assert xyz2 is None and torch.equal(kp2, kp0)
        assert torch.equal(img2, ops.convert_joints(torch.clamp(kp0, 0.0, 176.0), box, has, None))   # (boxes lie inside the frame)
    with pytest.raises(RuntimeError, match="no converted output|null"):
        from hn_amd._lib import check
        check(native.lib.hn_handnet_forward_xyz(native._h, rgb.data_ptr(), depth.data_ptr(), n, 480, 640, None, None, kp.data_ptr(),
                                                None, None, box.data_ptr(), has.data_ptr(), None), "hn_handnet_forward_xyz")


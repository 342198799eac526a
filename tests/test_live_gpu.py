"""SURVEY 8f #1 / #4 wired into the step: convert_joints + uvd2xyz in the aggregation's epilogue, the lifter's input on the
device, and the whole live chain (HandNet -> convert -> Pose2Mesh) as one captured step with one device -> host copy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PARAS = (617.343, 617.343, 312.42, 241.42)


def _heads(k, seed, peaked=None):
    """random A2J head tensors [k,11,11,336|672|336]; peaked = (uvd [k,21,3]): one dominant anchor per joint whose regression
    offset / depth put the aggregation's result EXACTLY on uvd (softmax weight 1.0 in fp32)."""
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn((k, 11, 11, 336), generator=g)
    reg = torch.randn((k, 11, 11, 672), generator=g) * 3.0
    dep = 0.3 + torch.rand((k, 11, 11, 336), generator=g)
    if peaked is not None:
        cls.fill_(-200.0)
        for i in range(k):
            for j in range(21):
                h, w, a = (i + j) % 11, (3 * j) % 11, j % 16
                c = a * 21 + j
                cls[i, h, w, c] = 50.0
                a0, a1 = h * 16 + 2 + 4 * (a // 4), w * 16 + 2 + 4 * (a % 4)
                reg[i, h, w, 2 * c] = float(peaked[i, j, 0]) - a0
                reg[i, h, w, 2 * c + 1] = float(peaked[i, j, 1]) - a1
                dep[i, h, w, c] = float(peaked[i, j, 2])
    return cls.cuda(), reg.cuda(), dep.cuda()


def test_aggregate_epilogue_equals_convert_joints_bit_for_bit():
    """hn_a2j_aggregate_convert_f32: crop uvd identical to hn_a2j_aggregate_f32's; image uvd and camera xyz identical to
    hn_convert_joints_f32 on it (one device function); zero rows for valid = 0, NaN rows for valid = 2; with the caller's
    clamps: identical to clamping first (torch.clamp) and converting then."""
    from hn_amd import ops
    k = 6
    cls, reg, dep = _heads(k, 11)
    box = torch.tensor([[0, 200, 49, 266], [192, 0, 264, 46], [100, 50, 420, 430], [0, 0, 640, 480], [7, 9, 8, 10], [500, 30, 640, 300]],
                       dtype=torch.int64).cuda()
    valid = torch.tensor([1, 1, 0, 1, 2, 1], dtype=torch.int32).cuda()
    plain = ops.a2j_aggregate(cls, reg, dep, valid=valid)
    uvd, img, xyz = ops.a2j_aggregate(cls, reg, dep, valid=valid, convert=dict(crop_box=box, paras=PARAS))
    assert torch.equal(uvd[valid == 1], plain[valid == 1]) and float(uvd[2].abs().max()) == 0.0 and bool(torch.isnan(uvd[4]).all())
    want_img = ops.convert_joints(plain, box, valid, None)
    want_xyz = ops.convert_joints(plain, box, valid, PARAS)
    ok = valid != 2
    assert torch.equal(img[ok], want_img[ok]) and torch.equal(xyz[ok], want_xyz[ok])
    assert bool(torch.isnan(img[4]).all()) and bool(torch.isnan(xyz[4]).all())
    assert float(img[2].abs().max()) == 0.0 and float(xyz[2].abs().max()) == 0.0
    # image uvd only (no intrinsics): no xyz
    uvd2, img2, xyz2 = ops.a2j_aggregate(cls, reg, dep, valid=valid, convert=dict(crop_box=box))
    assert xyz2 is None and torch.equal(img2[ok], want_img[ok]) and torch.equal(uvd2[ok], uvd[ok])
    # the live caller's clamps (ros_demo.py:279-283): keypoints to [0, 176], box x1,y1 to [0, H], x2,y2 to [0, W] -- on heads
    # whose dominant anchors put joints outside the crop, and boxes beyond the frame
    g = torch.Generator().manual_seed(5)
    pred = torch.rand((k, 21, 3), generator=g) * torch.tensor([240.0, 240.0, 1.2]) - torch.tensor([30.0, 30.0, -0.3])
    cls, reg, dep = _heads(k, 13, peaked=pred)
    box = torch.tensor([[0, 200, 49, 266], [500, 0, 700, 46], [100, 50, 420, 430], [-5, -3, 640, 480], [7, 9, 8, 10], [500, 490, 660, 500]],
                       dtype=torch.int64).cuda()
    plain = ops.a2j_aggregate(cls, reg, dep, valid=valid)
    uvd3, img3, xyz3 = ops.a2j_aggregate(cls, reg, dep, valid=valid,
                                         convert=dict(crop_box=box, paras=PARAS, clamp_keypoints=True, clamp_box=(480, 640)))
    kp_c = torch.clamp(plain, min=0.0, max=176.0)
    box_c = box.clone()
    box_c[:, :2] = torch.clamp(box_c[:, :2], 0, 480)
    box_c[:, 2:] = torch.clamp(box_c[:, 2:], 0, 640)
    assert not torch.equal(kp_c[ok], plain[ok]) and not torch.equal(box_c, box)        # (the case exercises both clamps)
    assert torch.equal(uvd3[ok], plain[ok])                                           # the network's own output is untouched
    assert torch.equal(img3[ok], ops.convert_joints(kp_c, box_c, valid, None)[ok])
    assert torch.equal(xyz3[ok], ops.convert_joints(kp_c, box_c, valid, PARAS)[ok])


def test_aggregate_epilogue_matches_reference_golden(golden_dir):
    """The reference's own convert_joints + uvd2xyz outputs (tests/golden/make_golden_joints.py) THROUGH the fused path: heads
    with one dominant anchor per joint make the aggregation return the golden `pred` exactly; its epilogue must then give
    the golden image uvd / camera xyz."""
    from hn_amd import ops
    g = np.load(golden_dir / "convert_joints.npz")
    pred = torch.from_numpy(g["pred"]).float()
    k = pred.shape[0]
    cls, reg, dep = _heads(k, 12, peaked=pred)
    box = torch.from_numpy(g["box"]).to(torch.int64).cuda()
    for i in range(k):      # (the fixture changes the camera from sample to sample: one launch per sample)
        uvd, img, xyz = ops.a2j_aggregate(cls[i:i + 1].contiguous(), reg[i:i + 1].contiguous(), dep[i:i + 1].contiguous(),
                                          convert=dict(crop_box=box[i:i + 1].contiguous(), paras=tuple(g["paras"][i])))
        assert (uvd.cpu()[0] - pred[i]).abs().max().item() < 2e-5
        assert np.abs(img.cpu().numpy()[0] - g["uvd_img"][i]).max() < 2e-4
        assert np.abs(xyz.cpu().numpy()[0] - g["xyz_pred"][i]).max() < 5e-3     # mm


def test_lifter_input_kernel_matches_the_callers_chain():
    """hn_joints2d_standardize_f32 vs oracle.pose2mesh_ref.lifter_input, the function-by-function restatement of
    ros_demo.py:148-157 (get_bbox -> process_bbox -> j2d_processing -> / input_shape -> (x - mean) / std)."""
    from hn_amd import ops
    from oracle import pose2mesh_ref
    rng = np.random.default_rng(4)
    n = 9
    uvd = np.zeros((n, 21, 3), dtype=np.float32)
    for i in range(n):
        uvd[i, :, :2] = rng.uniform(60, 560, 2) + rng.normal(size=(21, 2)) * rng.uniform(4, 90, 2)
        uvd[i, :, 2] = rng.uniform(0.3, 1.5, 21)
    valid = torch.tensor([1, 1, 1, 0, 1, 1, 2, 1, 1], dtype=torch.int32)
    got = ops.joints2d_standardize(torch.from_numpy(uvd).cuda(), valid.cuda()).cpu().numpy()
    for i in range(n):
        if int(valid[i]) != 1:
            assert not got[i].any()
            continue
        want = pose2mesh_ref.lifter_input(uvd[i, :, :2])
        assert np.abs(got[i] - want).max() < 3e-5, i


@pytest.fixture(scope="module")
def live(golden_dir, fcos_sd, a2j_sd):
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.live import LiveHandEngine
    from hn_amd.pipeline import HandNetEngine
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    from oracle import pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    p2m_sd = synth.make_pose2mesh_state_dict(seed=int(g["weight_seed"]), graph_sizes=[m.shape[0] for m in graphs])
    hand = HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    lifter = Pose2MeshEngine(p2m_sd, graphs, device="cuda")
    return LiveHandEngine(hand, lifter, PARAS, clamp=True), graphs, p2m_sd


def test_live_step_equals_its_parts_and_the_oracle_chain(live, fcos_sd, a2j_sd):
    """One live step (HandNet -> clamp + convert in the aggregation -> lifter input -> Pose2Mesh -> one copy) against (a) the
    same stages called one by one on the device: identical; (b) the oracle's chain on the CPU -- handnet_ref, the caller's
    numpy glue (clamp, convert_joints, lifter_input), pose2mesh_ref: crop boxes identical, joints within the pipeline's
    tolerance, mesh vertices within 2e-3 (the lifter's input is standardised: O(1) values, 1e-5 apart)."""
    from hn_amd import ops, synth
    from oracle import a2j_ref, handnet_ref, pose2mesh_ref
    eng, graphs, p2m_sd = live
    n = 3
    rgb, depth = synth.make_rgb(n, seed=1000), synth.make_depth(n, seed=2000)
    out = eng.forward_device(rgb.cuda(), depth.cuda())
    torch.cuda.synchronize()
    kp, has, box, words, (img, xyz), mesh = out.read()
    assert int((has == 1).sum()) == n and not any(words[:3])
    assert torch.equal(kp, out.hand.keypoints.cpu()) and torch.equal(img, out.hand.image_uvd.cpu()) and torch.equal(xyz, out.hand.xyz_mm.cpu())
    assert torch.equal(mesh, out.mesh.cpu()) and torch.equal(box, out.hand.crop_box.cpu())
    # (a) the parts, one by one
    kp_c = torch.clamp(out.hand.keypoints, 0.0, 176.0)
    assert torch.equal(out.hand.image_uvd, ops.convert_joints(kp_c, out.hand.crop_box, out.hand.has_hand, None))
    assert torch.equal(out.hand.xyz_mm, ops.convert_joints(kp_c, out.hand.crop_box, out.hand.has_hand, PARAS))
    p2d = ops.joints2d_standardize(out.hand.image_uvd, out.hand.has_hand)
    assert torch.equal(p2d, out.pose2d)
    mesh_parts, pose3d_parts = eng.lifter.forward(p2d)
    assert torch.equal(mesh_parts, out.mesh) and torch.equal(pose3d_parts, out.pose3d)
    # (b) the oracle's chain
    o_kp, _o_depth, o_crops = handnet_ref.handnet_forward([rgb[i] for i in range(n)], depth, fcos_sd, a2j_sd, 3)
    assert torch.equal(o_crops, box)
    assert (o_kp - kp).abs().max().item() < 1e-3
    for i in range(n):
        det = o_crops[i].clone()
        det[:2] = torch.clamp(det[:2], 0, 480)
        det[2:] = torch.clamp(det[2:], 0, 640)
        k = torch.clamp(o_kp[i], min=0.0, max=176.0).numpy()
        j2d = a2j_ref.convert_joints(k, det.numpy(), None)[:, :2]
        j3d = a2j_ref.convert_joints(k, det.numpy(), PARAS)
        assert np.abs(img[i].numpy()[:, :2] - j2d).max() < 2e-3 and np.abs(xyz[i].numpy() - j3d).max() < 2e-2
        x = pose2mesh_ref.lifter_input(j2d)
        o_mesh, _o_pose = pose2mesh_ref.pose2mesh_forward(torch.from_numpy(x)[None], p2m_sd, graphs)
        assert (o_mesh[0] - mesh[i]).abs().max().item() < 2e-3, i


def test_live_step_replays_from_one_graph(live):
    """The captured live step (one hipGraph: every launch of HandNet, the lifter and the copy) reproduces the eager step for
    new frames: integers bit-exact, floats to the split-K summation-order difference of capture-mode plans."""
    from hn_amd import synth
    eng, _, _ = live
    rgb, depth = synth.make_rgb(1, seed=1000).cuda(), synth.make_depth(1, seed=2000).cuda()
    run, s_img, s_dep, g_out = eng.graphed(rgb, depth)
    rgb2, depth2 = synth.make_rgb(1, seed=77).cuda(), synth.make_depth(1, seed=78).cuda()
    s_img.copy_(rgb2)
    s_dep.copy_(depth2)
    run()
    torch.cuda.synchronize()
    kp, has, box, words, (img, xyz), mesh = g_out.read()
    e = eng.forward_device(rgb2, depth2)
    torch.cuda.synchronize()
    e_kp, e_has, e_box, _w, (e_img, e_xyz), e_mesh = e.read()
    assert torch.equal(box, e_box) and torch.equal(has, e_has)
    assert (kp - e_kp).abs().max().item() < 2.5e-4 and (xyz - e_xyz).abs().max().item() < 5e-3
    assert (mesh - e_mesh).abs().max().item() < 1e-3


def test_dropin_set_convert_carries_the_joints_in_the_same_record(fcos_sd, a2j_sd):
    """HandNet.set_convert(paras): forward()'s tuple is unchanged and `last_converted` holds what the caller's
    convert_joints calls would compute from it (a2j.a2j.convert_joints, the numpy drop-in), eager and replayed."""
    import types
    from a2j.a2j import convert_joints
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    rgb, depth = synth.make_rgb(2, seed=1000).cuda(), synth.make_depth(2, seed=2000).cuda()
    images = [rgb[i] for i in range(2)]
    with torch.inference_mode():
        plain = net(images, depth_images=depth)
        net.set_convert(PARAS)
        for _ in range(6):          # (the fifth same-shape call replays a captured step)
            kp, depth_batch, crops = net(images, depth_images=depth)
            conv = net.last_converted
            assert torch.equal(kp, plain[0]) and torch.equal(crops, plain[2]) and torch.equal(depth_batch, plain[1])
            for i in range(2):
                want2d = convert_joints(kp[i].numpy(), None, crops[i].cpu().numpy(), None, 176, 176)
                want3d = convert_joints(kp[i].numpy(), None, crops[i].cpu().numpy(), np.asarray(PARAS), 176, 176)
                assert np.abs(conv["image_uvd"][i].numpy() - want2d).max() < 1e-3
                assert np.abs(conv["xyz_mm"][i].numpy() - want3d).max() < 2e-2


def test_live_demo_example_runs():
    """examples/live_demo.py: the reference's live loop (ros_demo.py:260-337) on the drop-in tree, three ways."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(repo / "examples" / "live_demo.py"), "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1. HandNet.forward: (1, 21, 3) cpu" in r.stdout and "2. set_convert" in r.stdout and "3. live step: 3 frames" in r.stdout
    assert "mesh (1, 1152, 3) -> (778, 3) camera-frame vertices; has_hand = [1]" in r.stdout


def test_converting_step_on_the_sparse_path_and_in_wide_records(fcos_sd, a2j_sd, monkeypatch):
    """set_convert on a SPARSE batch (A2J compacted to the frames with a hand): image uvd / camera xyz of the frames with a hand
    equal hn_convert_joints_f32 on the step's keypoints bit for bit, rows of frames without a hand are zeros in all three
    fields; and a to_host step carries all of it in ONE wide record (800 bytes per frame), row N = the range words."""
    from hn_amd import ops, pipeline, synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    eng = pipeline.HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3)
    eng.set_convert(PARAS, clamp=False)
    n = 16
    rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
    keep = torch.zeros((n,), dtype=torch.int32, device="cuda")
    keep[[2, 7, 12]] = 1
    real_crop, calls = ops.crop_resize, []

    def sparse_crop(*a, **k):
        box, has, crops = real_crop(*a, **k)
        return box * keep[:, None].to(box.dtype), has * keep, crops * keep[:, None, None, None].to(crops.dtype)
    monkeypatch.setattr(pipeline.ops, "crop_resize", sparse_crop)
    real_fwd = eng.a2j.forward_nhwc

    def spy(x, valid=None, **k):
        calls.append(x.shape[0])
        return real_fwd(x, valid=valid, **k)
    monkeypatch.setattr(eng.a2j, "forward_nhwc", spy)
    first = eng.forward_device(rgb, depth)                    # masked full batch; arms the hint
    torch.cuda.synchronize()
    second = eng.forward_device(rgb, depth, to_host=True)     # compacted, wide record
    torch.cuda.synchronize()
    assert calls == [n, 3], calls
    sel = keep.bool()
    for out in (first, second):
        assert torch.equal(out.image_uvd, ops.convert_joints(out.keypoints, out.crop_box, out.has_hand, None))
        assert torch.equal(out.xyz_mm, ops.convert_joints(out.keypoints, out.crop_box, out.has_hand, PARAS))
        for t in (out.keypoints, out.image_uvd, out.xyz_mm):
            assert float(t[~sel].abs().max()) == 0.0 and float(t[sel].abs().min()) > 0.0
    assert (second.keypoints[sel] - first.keypoints[sel]).abs().max().item() < 1e-4
    rec = second.host_record
    assert tuple(rec.shape) == (n + 1, 800)
    kp, has, box, words, more = pipeline.read_host_record(rec, n, extras=True)
    assert torch.equal(kp, second.keypoints.cpu()) and torch.equal(has, second.has_hand.cpu()) and torch.equal(box, second.crop_box.cpu())
    assert torch.equal(more[0], second.image_uvd.cpu()) and torch.equal(more[1], second.xyz_mm.cpu()) and not any(words[:3])
    # image uvd only (no intrinsics): 544-byte records
    eng.set_convert(None)
    third = eng.forward_device(rgb, depth, to_host=True)
    torch.cuda.synchronize()
    assert third.xyz_mm is None and tuple(third.host_record.shape) == (n + 1, 544)
    assert torch.equal(pipeline.read_host_record(third.host_record, n, extras=True)[4][0], third.image_uvd.cpu())


def test_live_forward_raw_equals_the_converted_feed(live):
    """LiveHandEngine.forward_raw (cv_bridge 'bgr8' + 16UC1 buffers from pageable host memory -> ingest kernel -> captured live
    step) against the same frames converted on the host as ros_demo.py:227-231,266-267 does and fed through graphed():
    identical records and mesh; two calls in a row with different frames (rotating staging buffers) stay apart."""
    eng, _, _ = live
    rng = np.random.default_rng(7)
    outs = []
    for i in range(2):
        bgr = rng.integers(0, 256, size=(1, 480, 640, 3), dtype=np.uint8)
        mm = rng.integers(300, 1500, size=(1, 480, 640)).astype(np.uint16)
        out = eng.forward_raw(torch.from_numpy(bgr), torch.from_numpy(mm))
        torch.cuda.synchronize()
        got = out.read()
        rgb = torch.from_numpy(bgr[..., ::-1].transpose(0, 3, 1, 2).astype(np.float32) / 255.0).cuda()
        dep = torch.from_numpy(mm.astype(np.float32) / 1000.0).unsqueeze(1).cuda()
        run, s_img, s_dep, ref = eng.graphed(rgb, dep)
        s_img.copy_(rgb)
        s_dep.copy_(dep)
        run()
        torch.cuda.synchronize()
        want = ref.read()
        assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]) and torch.equal(got[5], want[5])
        assert torch.equal(got[4][0], want[4][0]) and torch.equal(got[4][1], want[4][1])
        outs.append(got)
    assert not torch.equal(outs[0][0], outs[1][0])      # (different frames: different keypoints)


def test_dropin_results_at_batch_1_do_not_alias_the_step_record(fcos_sd, a2j_sd):
    """The keypoints HandNet.forward returns are the caller's to keep (handnet_pipeline.py:113-116 builds fresh tensors): the
    next call -- eager or replayed -- must not change them.  At ONE frame per call (ros_demo.py:270) a one-row slice of the
    pinned record is contiguous as it stands, and round 5's reader handed it back as a view: found by
    test_live_forward_raw_equals_the_converted_feed."""
    import types
    from handnet_pipeline.handnet_pipeline import HandNet
    from hn_amd import synth
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j="-"), num_classes=3)
    net.detector.load_state_dict(fcos_sd, strict=False)
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    frames = [(synth.make_rgb(1, seed=1000 + i).cuda(), synth.make_depth(1, seed=2000 + i).cuda()) for i in range(8)]
    kept = []
    with torch.inference_mode():
        for rgb, dep in frames:                      # (calls 5.. replay a captured step)
            kp, depth_batch, crops = net([rgb[0]], depth_images=dep)
            kept.append((kp, kp.clone(), depth_batch, depth_batch.clone(), crops, crops.clone()))
    assert net.engine().graph_count() == 1
    for kp, kp0, db, db0, cr, cr0 in kept:
        assert torch.equal(kp, kp0) and torch.equal(db, db0) and torch.equal(cr, cr0)
    assert not torch.equal(kept[0][0], kept[1][0]) and not torch.equal(kept[5][0], kept[6][0])


def test_live_step_hands_over_the_callers_final_mesh(live, golden_dir):
    """perm_reverse given, the live step also runs the caller's last lines (ros_demo.py:162,332-337): `mesh` of its outputs is
    out['mesh'] -- numpy float32 arithmetic on the raw vertices and the step's own joints3d, bit for bit."""
    from hn_amd import synth
    from hn_amd.live import LiveHandEngine
    eng, _, _ = live
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    perm = g["perm_reverse"][:778]
    rgb, depth = synth.make_rgb(2, seed=1000).cuda(), synth.make_depth(2, seed=2000).cuda()
    fin = LiveHandEngine(eng.hand, eng.lifter, PARAS, clamp=True, perm_reverse=perm)
    for graphed in (False, True):
        if graphed:
            run, s_img, s_dep, out = fin.graphed(rgb, depth)
            run()
        else:
            out = fin.forward_device(rgb, depth)
        torch.cuda.synchronize()
        kp, has, box, words, (img, xyz_rec), mesh = out.read()
        raw_mesh, xyz = out.raw_mesh.cpu().numpy(), out.hand.xyz_mm.cpu().numpy()     # (this step's own: a captured step may
        assert tuple(mesh.shape) == (2, 778, 3) and torch.equal(xyz_rec, out.hand.xyz_mm.cpu())   # plan its split-K otherwise)
        assert tuple(raw_mesh.shape) == (2, 1152, 3)
        for i in range(2):
            want = raw_mesh[i][perm, :]                              # pred_mesh[:, graph_perm_reverse[:V]]      (:162)
            want = want * 1000. + xyz[i][0]                          # out['mesh'] * 1000. + joints3d[0]         (:332)
            want /= 1000.                                            #                                           (:333)
            want[:, 1] *= -1                                         #                                           (:334)
            want[:, 2] *= -1                                         #                                           (:335)
            assert want.dtype == np.float32 and np.array_equal(mesh[i].numpy(), want)


@pytest.mark.parametrize("seed", [1, 2])
def test_live_chain_parity_on_other_weight_draws(golden_dir, seed):
    """The live chain against the oracle's chain (handnet_ref -> the caller's numpy glue -> pose2mesh_ref) AWAY from the seed-0
    weights: other draws of the detector, the pose network AND the lifter, structured frames beside noise -- crop boxes
    identical, joints within the pipeline's tolerance, final camera-frame mesh (out['mesh'], ros_demo.py:332-337) within 3e-3."""
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.live import LiveHandEngine
    from hn_amd.pipeline import HandNetEngine
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    from oracle import a2j_ref, handnet_ref, pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    perm = g["perm_reverse"][:778]
    fcos_sd, a2j_sd = synth.make_fcos_state_dict(seed, 3), synth.make_a2j_state_dict(seed)
    p2m_sd = synth.make_pose2mesh_state_dict(seed=seed + 10, graph_sizes=[m.shape[0] for m in graphs])
    eng = LiveHandEngine(HandNetEngine(FCOSEngine(fcos_sd, 3, device="cuda"), A2JEngine(a2j_sd, device="cuda"), 3),
                         Pose2MeshEngine(p2m_sd, graphs, device="cuda"), PARAS, clamp=True, perm_reverse=perm)
    rgb, depth = synth.make_rgb(2, seed=1200 + seed), synth.make_depth(2, seed=2200 + seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 480), torch.linspace(0, 1, 640), indexing="ij")
    rgb[1] = torch.stack([xx, yy, 0.5 * (xx + yy)])                      # a ramp frame beside the noise frame
    out = eng.forward_device(rgb.cuda(), depth.cuda())
    torch.cuda.synchronize()
    kp, has, box, words, (img, xyz), mesh = out.read()
    assert not any(words[:3])
    o_kp, _o_depth, o_crops = handnet_ref.handnet_forward([rgb[i] for i in range(2)], depth, fcos_sd, a2j_sd, 3)
    hands = [i for i in range(2) if int(has[i]) == 1]
    assert hands, "no frame with a hand in this case"
    if len(hands) == 2:
        assert torch.equal(o_crops, box)
    row = 0
    for i in hands:
        det = o_crops[row].clone()
        assert torch.equal(det, box[i])
        det[:2] = torch.clamp(det[:2], 0, 480)
        det[2:] = torch.clamp(det[2:], 0, 640)
        k = torch.clamp(o_kp[i], min=0.0, max=176.0).numpy()
        assert np.abs(k - torch.clamp(kp[i], 0.0, 176.0).numpy()).max() < 1e-3
        j2d = a2j_ref.convert_joints(k, det.numpy(), None)[:, :2]
        j3d = a2j_ref.convert_joints(k, det.numpy(), PARAS)
        x = pose2mesh_ref.lifter_input(j2d)
        o_mesh, _ = pose2mesh_ref.pose2mesh_forward(torch.from_numpy(x)[None], p2m_sd, graphs)
        want = o_mesh[0].numpy()[perm, :] * 1000. + j3d[0]
        want /= 1000.
        want[:, 1] *= -1
        want[:, 2] *= -1
        assert np.abs(mesh[i].numpy() - want).max() < 3e-3, (seed, i, np.abs(mesh[i].numpy() - want).max())
        row += 1


def test_live_step_at_batch_32_is_frame_independent(live):
    """BASELINE config 4's batch through the live step: permuting the 32 frames permutes every output (crop boxes, flags,
    keypoints, converted joints, mesh vertices) bit for bit -- the detector, the crop, A2J, the fused conversion and the
    lifter's layer-by-layer form above 4 samples keep frames apart -- and frame 0 agrees with the single-frame step (the
    lifter's fused latency form, other split-K plans) to the summation-order tolerance."""
    from hn_amd import synth
    eng, _, _ = live
    n = 32
    rgb, depth = synth.make_rgb(n, seed=1000), synth.make_depth(n, seed=2000)
    out = eng.forward_device(rgb.cuda(), depth.cuda())
    torch.cuda.synchronize()
    kp, has, box, words, (img, xyz), mesh = out.read()
    assert not any(words[:3]) and int((has == 1).sum()) == n and bool(torch.isfinite(mesh).all())
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(9))
    out_p = eng.forward_device(rgb[perm].cuda(), depth[perm].cuda())
    torch.cuda.synchronize()
    kp_p, has_p, box_p, _w, (img_p, xyz_p), mesh_p = out_p.read()
    assert torch.equal(box_p, box[perm]) and torch.equal(has_p, has[perm]) and torch.equal(kp_p, kp[perm])
    assert torch.equal(img_p, img[perm]) and torch.equal(xyz_p, xyz[perm]) and torch.equal(mesh_p, mesh[perm])
    one = eng.forward_device(rgb[:1].cuda(), depth[:1].cuda())
    torch.cuda.synchronize()
    kp1, _h1, box1, _w1, (_img1, xyz1), mesh1 = one.read()
    assert torch.equal(box1[0], box[0]) and (kp1[0] - kp[0]).abs().max().item() < 2.5e-4
    assert (xyz1[0] - xyz[0]).abs().max().item() < 5e-3 and (mesh1[0] - mesh[0]).abs().max().item() < 1e-3

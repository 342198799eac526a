"""SURVEY 8f #1's second caller: the evaluation loop (A2JModelLightning.test_step, a2j/a2j.py:333-359) -- forward -> convert_joints
on the prediction AND the ground truth with the DATASET's float32 box and each sample's own intrinsics
(datasets3d/a2jdataset.py:262-265,279,293) -> RMSE in mm -> the HPE evaluator's text file.  Goldens: the reference's own
convert_joints outputs on float32 operands (tests/golden/make_golden_joints.py, `*_f32` arrays): every operation stays in fp32
there, so the device path is compared BIT FOR BIT."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_samples_conversion_is_bit_identical_to_the_reference_golden(golden_dir):
    """hn_convert_joints_samples_f32 (fp32 boxes with fractional corners, one camera per sample) == the imported reference's
    convert_joints + uvd2xyz on the same float32 operands, all 12 cases in ONE launch: torch.equal, prediction and ground truth,
    image uvd and camera xyz; rows with valid = 0 are zeros."""
    from hn_amd import ops
    g = np.load(golden_dir / "convert_joints.npz")
    pred, gt = torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["gt"]).cuda()
    box, paras = torch.from_numpy(g["box_f32"]).cuda(), torch.from_numpy(g["paras"]).cuda()
    img, xyz = ops.convert_joints_samples(pred, box, paras)
    assert torch.equal(img.cpu(), torch.from_numpy(g["uvd_img_f32"])) and torch.equal(xyz.cpu(), torch.from_numpy(g["xyz_pred_f32"]))
    none, xyz_gt = ops.convert_joints_samples(gt, box, paras, want_image=False)
    assert none is None and torch.equal(xyz_gt.cpu(), torch.from_numpy(g["xyz_gt_f32"]))
    img_only, no_xyz = ops.convert_joints_samples(pred, box)
    assert no_xyz is None and torch.equal(img_only, img)
    valid = torch.ones((pred.shape[0],), dtype=torch.int32)
    valid[3] = 0
    img_v, xyz_v = ops.convert_joints_samples(pred, box, paras, valid=valid.cuda())
    keep = valid.bool()
    assert torch.equal(img_v.cpu()[keep], img.cpu()[keep]) and torch.equal(xyz_v.cpu()[keep], xyz.cpu()[keep])
    assert not img_v[3].any() and not xyz_v[3].any()
    with pytest.raises(ValueError):
        ops.convert_joints_samples(pred, box[:5].contiguous(), paras)
    with pytest.raises(ValueError):
        ops.convert_joints_samples(pred, box, want_image=False)


def test_aggregation_epilogue_takes_the_datasets_operands(golden_dir):
    """The fused form (hn_a2j_aggregate_convert_f32 with opts->sample_box / sample_paras): heads with one dominant anchor per
    joint return the golden `pred`; the epilogue's image uvd / camera xyz are bit-identical to the stand-alone kernel on the
    aggregation's own output (one device function) and within float rounding of `pred`'s goldens."""
    from hn_amd import ops
    from test_live_gpu import _heads
    g = np.load(golden_dir / "convert_joints.npz")
    pred = torch.from_numpy(g["pred"]).float()
    k = pred.shape[0]
    cls, reg, dep = _heads(k, 12, peaked=pred)
    box, paras = torch.from_numpy(g["box_f32"]).cuda(), torch.from_numpy(g["paras"]).cuda()
    uvd, img, xyz = ops.a2j_aggregate(cls, reg, dep, convert=dict(sample_box=box, sample_paras=paras))
    plain = ops.a2j_aggregate(cls, reg, dep)
    assert torch.equal(uvd, plain)
    want_img, want_xyz = ops.convert_joints_samples(plain, box, paras)
    assert torch.equal(img, want_img) and torch.equal(xyz, want_xyz)
    assert np.abs(img.cpu().numpy() - g["uvd_img_f32"]).max() < 2e-4 and np.abs(xyz.cpu().numpy() - g["xyz_pred_f32"]).max() < 5e-3
    # one camera for all samples + the dataset's boxes; the detector's int64 boxes + one camera per sample
    p0 = tuple(float(v) for v in g["paras"][0])
    _, _, xyz_one = ops.a2j_aggregate(cls, reg, dep, convert=dict(sample_box=box, paras=p0))
    p0_rows = torch.from_numpy(np.tile(g["paras"][0], (k, 1))).cuda()
    assert torch.equal(xyz_one, ops.convert_joints_samples(plain, box, p0_rows)[1])
    ibox = torch.from_numpy(g["box"]).cuda()
    _, img_i, xyz_i = ops.a2j_aggregate(cls, reg, dep, convert=dict(crop_box=ibox, sample_paras=paras))
    assert torch.equal(img_i, ops.convert_joints(plain, ibox, None, None))
    assert torch.equal(xyz_i, ops.convert_joints_samples(plain, ibox.float().contiguous(), paras)[1])
    # rows without a crop are zeros, rows of a non-finite crop NaN -- in the converted outputs too
    valid = torch.ones((k,), dtype=torch.int32)
    valid[2], valid[5] = 0, 2
    uvd_v, img_v, xyz_v = ops.a2j_aggregate(cls, reg, dep, valid=valid.cuda(), convert=dict(sample_box=box, sample_paras=paras))
    ok = (valid == 1).cuda()
    assert torch.equal(img_v[ok], img[ok]) and torch.equal(xyz_v[ok], xyz[ok]) and torch.equal(uvd_v[ok], uvd[ok])
    assert not uvd_v[2].any() and not img_v[2].any() and not xyz_v[2].any()
    assert bool(torch.isnan(uvd_v[5]).all()) and bool(torch.isnan(img_v[5]).all()) and bool(torch.isnan(xyz_v[5]).all())
    with pytest.raises(ValueError):
        ops.a2j_aggregate(cls, reg, dep, convert=dict(crop_box=ibox, sample_box=box))
    with pytest.raises(ValueError):
        ops.a2j_aggregate(cls, reg, dep, convert=dict(sample_box=box, paras=p0, sample_paras=paras))


def _eval_batch(k, seed):
    """What datasets3d/a2jdataset.py:293 hands the loop: (depth crop, gt uvd, id, colour crop, float32 box, float32 paras,
    combined)."""
    from hn_amd import synth
    g = torch.Generator().manual_seed(seed)
    im = synth.make_crops(k, 176, seed=seed)
    gt = torch.rand((k, 21, 3), generator=g) * torch.tensor([176.0, 176.0, 0.8]) + torch.tensor([0.0, 0.0, 0.4])
    x1, y1 = torch.rand((k,), generator=g) * 380, torch.rand((k,), generator=g) * 260
    box = torch.stack([x1, y1, x1 + 40 + torch.rand((k,), generator=g) * 200, y1 + 40 + torch.rand((k,), generator=g) * 170], dim=1)
    paras = torch.tensor([[617.343, 617.343, 312.42, 241.42]]).repeat(k, 1) + torch.rand((k, 4), generator=g) * 20
    ids = torch.arange(100 * seed, 100 * seed + k, dtype=torch.int64).reshape(k, 1)
    return im, gt, ids, None, box.float(), paras.float(), None


def test_lightning_test_step_is_the_references_evaluation_loop(tmp_path, a2j_sd):
    """A2JModelLightning.test_step on a batch of ONE (the reference's own evaluation batch: its convert_joints reshapes the
    whole batch against one box) and on a batch of five: the logged RMSE and the evaluator's text file against the oracle's
    chain -- a2j_ref.a2j_forward -> a2j_ref.convert_joints (fp32 operands) -> numpy RMSE -> the reference's line format --
    and forward_xyz against forward + the stand-alone conversion (bit-identical)."""
    from a2j.a2j import A2JModelLightning
    from hn_amd import ops
    from oracle import a2j_ref
    net = A2JModelLightning(output_dir=str(tmp_path / "out"))
    net.a2j.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    lines_want = []
    for step, k in enumerate((1, 5)):
        batch = _eval_batch(k, seed=7 + step)
        im, gt, ids, _, box, paras, _ = batch
        rmse = net.test_step(batch, step)
        o_kp = a2j_ref.a2j_forward(im, a2j_sd)
        pred = np.stack([a2j_ref.convert_joints(o_kp[i].numpy(), box[i].numpy(), paras[i].numpy()) for i in range(k)])
        want_gt = np.stack([a2j_ref.convert_joints(gt[i].numpy(), box[i].numpy(), paras[i].numpy()) for i in range(k)])
        want = np.sqrt(np.mean(np.square(want_gt.reshape(-1, 3) - pred.reshape(-1, 3))))
        assert abs(float(rmse) - float(want)) < 1e-4 * float(want) and net.logged["test_rmse"][step] == float(rmse)
        lines_want += [(int(ids[i, 0]), pred[i]) for i in range(k)]
        # the forward with the conversion in its epilogue == forward, then the stand-alone kernels
        kp, img, xyz = net.a2j.forward_xyz(im, box, paras)
        assert torch.equal(kp, net(im.cuda()))
        w_img, w_xyz = ops.convert_joints_samples(kp.cuda(), box.cuda(), paras.cuda())
        assert torch.equal(img, w_img.cpu()) and torch.equal(xyz, w_xyz.cpu())
        assert (kp - o_kp).abs().max().item() < 1e-3 and np.abs(xyz.numpy() - pred).max() < 0.02      # mm
    text = (tmp_path / "out" / "a2j_test_metrics" / "s0_test_0.txt").read_text().splitlines()
    assert len(text) == 6 and not any(" " in l for l in text)
    for line, (ident, pred) in zip(text, lines_want):
        cells = line.split(",")
        assert int(cells[0]) == ident and len(cells) == 1 + 63
        assert np.abs(np.array(cells[1:], dtype=np.float32).reshape(21, 3) - pred).max() < 0.02
    # the digits are float32's shortest round trip (what numpy 1.x prints inside a list): parsing them gives the value back
    kp, img, xyz = net.a2j.forward_xyz(*[_eval_batch(5, seed=8)[i] for i in (0, 4, 5)])
    assert np.array_equal(np.array(text[1].split(",")[1:], dtype=np.float32), xyz[0].numpy().reshape(-1))
    # one camera for the whole batch, the detector's int64 boxes (the stand-alone demos' form)
    im, _, _, _, box, paras, _ = _eval_batch(3, seed=11)
    ibox = box.round().to(torch.int64)
    kp, img, xyz = net.a2j.forward_xyz(im, ibox, paras[0])
    assert torch.equal(img, ops.convert_joints(kp.cuda(), ibox.cuda(), None, None).cpu())
    assert torch.equal(xyz, ops.convert_joints(kp.cuda(), ibox.cuda(), None, tuple(float(v) for v in paras[0])).cpu())
    assert net.a2j.forward_xyz(im, ibox)[2] is None
    with pytest.raises(NotImplementedError):
        net.validation_step(None, 0)
    with pytest.raises(ImportError):
        net.test_epoch_end([])


def test_crop_mesh_step_is_the_mesh_demos_loop_body(golden_dir, a2j_sd):
    """hn_amd.live.CropMeshEngine (a2j_mesh.py:58-80: dataset crops -> A2J -> np.clip -> convert_joints twice -> predict_mesh ->
    the final mesh) against (a) the same stages one by one on the device: identical; (b) the oracle's chain on the CPU: joints
    within the A2J tolerance, the final camera-frame mesh within 3e-3; and the captured step against the eager one."""
    from a2j.a2j import A2JModel
    from hn_amd import ops, synth
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    from oracle import a2j_ref, pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    perm = g["perm_reverse"][:778]
    p2m_sd = synth.make_pose2mesh_state_dict(seed=int(g["weight_seed"]), graph_sizes=[m.shape[0] for m in graphs])
    net = A2JModel(21, 176, 176)
    net.load_state_dict(a2j_sd, strict=False)
    net = net.cuda().eval()
    eng = net.mesh(Pose2MeshEngine(p2m_sd, graphs, device="cuda"), clamp=True, perm_reverse=perm)
    k = 3
    im, _gt, _ids, _, box, paras, _ = _eval_batch(k, seed=21)
    out = eng.forward_device(im.cuda(), box.cuda(), paras.cuda())
    torch.cuda.synchronize()
    kp, img, xyz, mesh, words = out.read()
    assert not any(words[:3]) and tuple(mesh.shape) == (k, 778, 3)
    assert torch.equal(kp, out.keypoints.cpu()) and torch.equal(mesh, out.mesh.cpu()) and torch.equal(xyz, out.xyz_mm.cpu())
    # (a) the parts
    assert torch.equal(kp, net(im.cuda()))
    kp_c = torch.clamp(out.keypoints, 0.0, 176.0)
    w_img, w_xyz = ops.convert_joints_samples(kp_c, box.cuda(), paras.cuda())
    assert torch.equal(out.image_uvd, w_img) and torch.equal(out.xyz_mm, w_xyz)
    p2d = ops.joints2d_standardize(w_img)
    raw, pose3d = eng.lifter.forward(p2d)
    assert torch.equal(p2d, out.pose2d) and torch.equal(raw, out.raw_mesh) and torch.equal(pose3d, out.pose3d)
    assert torch.equal(ops.mesh_finish(raw, eng.perm, w_xyz), out.mesh)
    # (b) the oracle's chain, line by line as a2j_mesh.py writes it
    o_kp = a2j_ref.a2j_forward(im, a2j_sd)
    assert (o_kp - kp).abs().max().item() < 1e-3
    for i in range(k):
        keypoint_pred = np.clip(o_kp[i].numpy(), a_min=0.0, a_max=176.0)
        joints2d = a2j_ref.convert_joints(keypoint_pred, box[i].numpy(), None)[:, :2]
        joints3d = a2j_ref.convert_joints(keypoint_pred, box[i].numpy(), paras[i].numpy())
        assert np.abs(img[i].numpy()[:, :2] - joints2d).max() < 2e-3 and np.abs(xyz[i].numpy() - joints3d).max() < 2e-2
        o_mesh, _ = pose2mesh_ref.pose2mesh_forward(torch.from_numpy(pose2mesh_ref.lifter_input(joints2d))[None], p2m_sd, graphs)
        want = o_mesh[0].numpy()[perm, :] * 1000. + joints3d[0]
        want /= 1000.
        want[:, 1] *= -1
        want[:, 2] *= -1
        assert np.abs(mesh[i].numpy() - want).max() < 3e-3, (i, np.abs(mesh[i].numpy() - want).max())
    # the captured step on a new batch == the eager step on it (floats to the split-K order of capture-mode plans)
    run, s_crops, s_box, s_paras, g_out = eng.graphed(im.cuda(), box.cuda(), paras.cuda())
    im2, _, _, _, box2, paras2, _ = _eval_batch(k, seed=22)
    s_crops.copy_(im2)
    s_box.copy_(box2)
    s_paras.copy_(paras2)
    run()
    torch.cuda.synchronize()
    g_kp, g_img, g_xyz, g_mesh, g_words = g_out.read()
    e = eng.forward_device(im2.cuda(), box2.cuda(), paras2.cuda())
    torch.cuda.synchronize()
    e_kp, e_img, e_xyz, e_mesh, _ = e.read()
    assert not any(g_words[:3]) and not torch.equal(g_kp, kp)
    assert (g_kp - e_kp).abs().max().item() < 2.5e-4 and (g_xyz - e_xyz).abs().max().item() < 5e-3
    assert (g_mesh - e_mesh).abs().max().item() < 1e-3
    # raw vertices without perm_reverse; a crop with a NaN pixel gives NaN rows, not a wrong mesh
    plain = net.mesh(eng.lifter, clamp=True)
    bad = im.clone()
    bad[1, 0, 5, 5] = float("nan")
    o2 = plain.forward_device(bad.cuda(), box.cuda(), paras.cuda())
    torch.cuda.synchronize()
    kp2, _img2, xyz2, mesh2, _ = o2.read()
    assert tuple(mesh2.shape) == (k, 1152, 3) and bool(torch.isnan(kp2[1]).all()) and bool(torch.isnan(xyz2[1]).all())
    assert torch.equal(kp2[0], kp[0]) and torch.equal(mesh2[0], out.raw_mesh.cpu()[0])


def test_crop_mesh_step_at_batch_64_is_row_independent(golden_dir, a2j_sd):
    """BASELINE config 2's batch (64 crops) through the mesh demo's step: a row's results depend on its own crop, box and
    intrinsics only -- permuting the batch permutes every output bit for bit (A2J at 64 crops, the fused conversion, the
    lifter's layer-by-layer form above 4 samples, the mesh finish) -- and the first rows agree with a batch of 3 (the
    lifter's fused latency form) to the summation-order difference of the two lifter forms."""
    from a2j.a2j import A2JModel
    from hn_amd import synth
    from hn_amd.pose2mesh_engine import Pose2MeshEngine
    from oracle import pose2mesh_ref
    g = np.load(golden_dir / "pose2mesh_forward.npz")
    graphs = pose2mesh_ref.load_graphs(g)
    p2m_sd = synth.make_pose2mesh_state_dict(seed=int(g["weight_seed"]), graph_sizes=[m.shape[0] for m in graphs])
    net = A2JModel(21, 176, 176)
    net.load_state_dict(a2j_sd, strict=False)
    eng = net.cuda().eval().mesh(Pose2MeshEngine(p2m_sd, graphs, device="cuda"), clamp=True, perm_reverse=g["perm_reverse"][:778])
    k = 64
    im, _gt, _ids, _, box, paras, _ = _eval_batch(k, seed=31)
    out = eng.forward_device(im.cuda(), box.cuda(), paras.cuda())
    torch.cuda.synchronize()
    kp, img, xyz, mesh, words = out.read()
    assert not any(words[:3]) and tuple(mesh.shape) == (k, 778, 3) and bool(torch.isfinite(mesh).all())
    perm = torch.randperm(k, generator=torch.Generator().manual_seed(5))
    out_p = eng.forward_device(im[perm].cuda(), box[perm].cuda(), paras[perm].cuda())
    torch.cuda.synchronize()
    kp_p, img_p, xyz_p, mesh_p, _ = out_p.read()
    assert torch.equal(kp_p, kp[perm]) and torch.equal(img_p, img[perm]) and torch.equal(xyz_p, xyz[perm])
    assert torch.equal(mesh_p, mesh[perm])
    small = eng.forward_device(im[:3].cuda(), box[:3].cuda(), paras[:3].cuda())
    torch.cuda.synchronize()
    kp_s, _img_s, xyz_s, mesh_s, _ = small.read()
    assert (kp_s - kp[:3]).abs().max().item() < 2.5e-4 and (xyz_s - xyz[:3]).abs().max().item() < 5e-3
    assert (mesh_s - mesh[:3]).abs().max().item() < 1e-3

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

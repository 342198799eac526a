"""The drop-in packages must satisfy the import surface of the reference's callers (no GPU needed): every
`from <module> import <symbol>` those scripts execute at module load for the hot path resolves from
handnet-pipeline_amd/, and the objects have the constructor / classmethod contract the callers use."""
import importlib
import inspect

import pytest
import torch

# (module, symbol, reference call site) -- data, not code: what ros_demo.py / a2j_infer.py / a2j_mesh.py /
# handnet_pipeline.py import from the packages this repo replaces
CALLER_IMPORTS = [
    ("handnet_pipeline.handnet_pipeline", "HandNet", "ros_demo.py:10"),
    ("a2j.a2j", "convert_joints", "ros_demo.py:38"),
    ("a2j.a2j", "A2JModel", "a2j_infer.py:3"),
    ("a2j.a2j", "A2JModelLightning", "a2j_infer.py:12"),
    ("a2j.a2j", "convert_joints", "a2j_mesh.py:5"),
    ("a2j.a2j", "A2JModel", "a2j_mesh.py:5"),
    ("a2j.a2j", "A2JModelLightning", "a2j_mesh.py:15"),
    ("a2j.a2j", "A2JModel", "handnet_pipeline/handnet_pipeline.py:4"),
    ("fcos_utils.fcos", "FCOS", "handnet_pipeline/handnet_pipeline.py:7"),
    ("a2j.a2j", "A2JModelLightning", "handnet_pipeline/handnet_pipeline.py:8"),
    ("handnet_pipeline.handnet_pipeline", "load_pretrained_fcos", "handnet_pipeline/handnet_pipeline.py:14"),
    ("handnet_pipeline.handnet_pipeline", "load_pretrained_a2j", "handnet_pipeline/handnet_pipeline.py:25"),
    ("models.pose2mesh_net", "get_model", "ros_demo.py:30,148-168"),
]


@pytest.mark.parametrize("module,symbol,site", CALLER_IMPORTS)
def test_caller_import_resolves_from_dropin(module, symbol, site):
    import sys
    from pathlib import Path
    pkg = Path(__file__).resolve().parent.parent / "handnet-pipeline_amd"
    if module.startswith("models."):       # pose2mesh sits under pose2mesh/lib like in the reference (ros_demo.py:22-23)
        p = str(pkg / "pose2mesh" / "lib")
        if p not in sys.path:
            sys.path.insert(0, p)
    m = importlib.import_module(module)
    assert str(pkg) in str(Path(m.__file__).resolve()), f"{module} resolved outside the drop-in tree"
    assert hasattr(m, symbol), f"{site}: `from {module} import {symbol}` would raise ImportError"


def test_constructor_signatures_match_reference():
    from a2j.a2j import A2JModel, A2JModelLightning
    from fcos_utils.fcos import FCOS
    from handnet_pipeline.handnet_pipeline import HandNet, HandNetPipeline
    assert HandNetPipeline is HandNet
    names = lambda f: list(inspect.signature(f).parameters)[1:]  # noqa: E731
    assert names(HandNet.__init__) == ["args", "reload_detector", "num_classes", "reload_a2j", "RGBD"]
    assert names(HandNet.forward) == ["images", "depth_images", "is_3D", "is_detect"]
    assert names(A2JModel.__init__) == ["num_classes", "crop_height", "crop_width", "is_3D", "is_RGBD", "spatial_factor"]
    assert names(A2JModelLightning.__init__) == ["num_classes", "crop_height", "crop_width", "is_3D", "is_RGBD",
                                                 "spatial_factor", "display_freq", "output_dir"]
    assert names(FCOS.__init__)[:2] == ["num_classes", "ext"]
    assert inspect.signature(FCOS.__init__).parameters["ext"].default is True


def test_lightning_checkpoint_round_trip(tmp_path, a2j_rgbd_sd):
    """A2JModelLightning.load_from_checkpoint on a Lightning-layout file (state_dict keys `a2j.*`, constructor
    arguments under hyper_parameters): the RGBD stem width comes from the file, weights land in `.a2j`."""
    from a2j.a2j import A2JModel, A2JModelLightning
    path = tmp_path / "a2j_rgbd.ckpt"
    torch.save({"state_dict": {"a2j." + k: v for k, v in a2j_rgbd_sd.items()},
                "hyper_parameters": {"num_classes": 21, "crop_height": 176, "crop_width": 176, "is_3D": True,
                                     "is_RGBD": True, "spatial_factor": 0.5, "display_freq": 5000,
                                     "output_dir": "models/a2j"}}, path)
    m = A2JModelLightning.load_from_checkpoint(str(path)).eval()
    assert isinstance(m.a2j, A2JModel) and m.rgbd and m.a2j.is_RGBD
    got = m.state_dict()
    assert all(k.startswith("a2j.") for k in got)
    k = "a2j.Backbone.model.conv1.weight"
    assert got[k].shape[1] == 4 and torch.equal(got[k], a2j_rgbd_sd[k[4:]])
    # a real Lightning file also carries the anchor / threshold buffers of A2J_loss and post_process: ignored;
    # a file that lacks one of OUR weights fails under strict (Lightning's default)
    extra = torch.load(path)
    extra["state_dict"]["a2j.post_process.all_anchors"] = torch.zeros(1936, 2)
    extra["state_dict"]["a2j.criterion.thres"] = torch.tensor([16.0, 32.0])
    torch.save(extra, tmp_path / "extra.ckpt")
    assert A2JModelLightning.load_from_checkpoint(str(tmp_path / "extra.ckpt")).rgbd
    del extra["state_dict"]["a2j.regressionModel.output.weight"]
    torch.save(extra, tmp_path / "short.ckpt")
    with pytest.raises(KeyError, match="regressionModel.output.weight"):
        A2JModelLightning.load_from_checkpoint(str(tmp_path / "short.ckpt"))
    assert A2JModelLightning.load_from_checkpoint(str(tmp_path / "short.ckpt"), strict=False).rgbd
    with pytest.raises(NotImplementedError):
        m.training_step(None, 0)
    with pytest.raises(KeyError):      # a plain {"model": sd} file is not a Lightning checkpoint
        torch.save({"model": a2j_rgbd_sd}, tmp_path / "plain.pth")
        A2JModelLightning.load_from_checkpoint(str(tmp_path / "plain.pth"))


def test_handnet_ckpt_path_goes_through_lightning(tmp_path, a2j_sd):
    """handnet_pipeline.py:27-29: 'ckpt' in the path -> A2JModelLightning.load_from_checkpoint(...).eval()."""
    import types
    from a2j.a2j import A2JModelLightning
    from handnet_pipeline.handnet_pipeline import HandNet
    path = tmp_path / "a2j.ckpt"
    torch.save({"state_dict": {"a2j." + k: v for k, v in a2j_sd.items()}, "hyper_parameters": {"is_RGBD": False}}, path)
    net = HandNet(types.SimpleNamespace(pretrained_fcos="-", pretrained_a2j=str(path)), num_classes=3)
    assert isinstance(net.a2j, A2JModelLightning) and net.RGBD is False
    assert torch.equal(net.a2j.a2j.state_dict()["regressionModel.output.bias"], a2j_sd["regressionModel.output.bias"])


def test_is_3d_false_fails_like_the_reference():
    """The reference's two-head model constructs but its forward dies unpacking two heads into three names
    (a2j/a2j.py:223 builds post_process with is_3D=True; a2j/anchor.py:58-59)."""
    from a2j.a2j import A2JModel
    m = A2JModel(21, 176, 176, is_3D=False)
    assert not any(k.startswith("DepthRegressionModel") for k in m.state_dict())
    with pytest.raises(ValueError, match="not enough values to unpack"):
        m(torch.zeros(1, 1, 176, 176))


def test_convert_joints_dropin_matches_reference_golden(golden_dir):
    """The numpy drop-in `a2j.a2j.convert_joints` and the oracle restatement against outputs of the reference's
    own function (tests/golden/make_golden_joints.py)."""
    import numpy as np
    from a2j.a2j import convert_joints
    from oracle import a2j_ref
    g = np.load(golden_dir / "convert_joints.npz")
    for i in range(g["pred"].shape[0]):
        a, b = convert_joints(g["pred"][i], g["gt"][i], g["box"][i], g["paras"][i], 176, 176)
        assert np.abs(a - g["xyz_pred"][i]).max() < 2e-3 and np.abs(b - g["xyz_gt"][i]).max() < 2e-3   # mm
        assert np.abs(convert_joints(g["pred"][i], None, g["box"][i], None, 176, 176) - g["uvd_img"][i]).max() < 1e-4
        assert np.abs(a2j_ref.convert_joints(g["pred"][i], g["box"][i], g["paras"][i]) - g["xyz_pred"][i]).max() < 2e-3
        assert np.abs(a2j_ref.convert_joints(g["pred"][i], g["box"][i], None) - g["uvd_img"][i]).max() < 1e-4
        # the evaluation caller's operands (a2j/a2j.py:339-346: the dataset's float32 box, fractional corners): every operation
        # stays in float32, so the drop-in and the oracle reproduce the reference BIT FOR BIT
        a, b = convert_joints(g["pred"][i], g["gt"][i], g["box_f32"][i], g["paras"][i], 176, 176)
        assert a.dtype == np.float32 and np.array_equal(a, g["xyz_pred_f32"][i]) and np.array_equal(b, g["xyz_gt_f32"][i])
        assert np.array_equal(convert_joints(g["pred"][i], None, g["box_f32"][i], None, 176, 176), g["uvd_img_f32"][i])
        assert np.array_equal(a2j_ref.convert_joints(g["pred"][i], g["box_f32"][i], g["paras"][i]), g["xyz_pred_f32"][i])
        assert np.array_equal(a2j_ref.convert_joints(g["gt"][i], g["box_f32"][i], g["paras"][i]), g["xyz_gt_f32"][i])
        assert np.array_equal(a2j_ref.convert_joints(g["pred"][i], g["box_f32"][i], None), g["uvd_img_f32"][i])


def test_weight_range_contract_is_checked_at_load():
    from hn_amd.weights import split_f16x3
    w = torch.randn(8, 3, 3, 32)
    split_f16x3(w)
    w[2, 1, 1, 5] = 7.0e4
    with pytest.raises(ValueError, match="fp16 range"):
        split_f16x3(w)
    w[2, 1, 1, 5] = float("nan")
    with pytest.raises(ValueError, match="fp16 range"):
        split_f16x3(w)


def test_wide_host_record_layout():
    """hn_amd.pipeline.read_host_record on a record buffer packed by hand: the 296-byte record of the step (crop box 4 x int64,
    has_hand, row flag, keypoints) and the WIDE record of a converting step (image uvd and camera xyz behind the keypoints), row
    N = the range words -- the layout hn_pack_records_ex writes and the live step's one copy carries."""
    import numpy as np
    from hn_amd.pipeline import RECORD_BYTES, read_host_record, record_bytes
    assert record_bytes(1) == RECORD_BYTES == 296 and record_bytes(2) == 544 and record_bytes(3) == 800
    n = 3
    rng = np.random.default_rng(0)
    kp, img, xyz = (rng.normal(size=(n, 21, 3)).astype(np.float32) for _ in range(3))
    box = rng.integers(0, 640, size=(n, 4)).astype(np.int64)
    has = np.array([1, 0, 2], dtype=np.int32)
    for fields, extras in ((1, ()), (3, (img, xyz))):
        rec = np.zeros((n + 1, record_bytes(fields)), dtype=np.uint8)
        rec[:n, :32] = box.view(np.uint8).reshape(n, 32)
        rec[:n, 32:36] = has.view(np.uint8).reshape(n, 4)
        rec[:n, 36:40] = np.ones(n, dtype=np.int32).view(np.uint8).reshape(n, 4)
        for f, t in enumerate((kp,) + tuple(extras)):
            rec[:n, 40 + 252 * f: 40 + 252 * (f + 1)] = t.reshape(n, 63).view(np.uint8)
        rec[n, :16] = np.array([0, 1, 0, 0], dtype=np.int32).view(np.uint8)
        out = read_host_record(torch.from_numpy(rec), n, extras=bool(extras))
        assert np.array_equal(out[0].numpy(), kp) and out[1].tolist() == [1, 0, 2] and np.array_equal(out[2].numpy(), box)
        assert out[3] == [0, 1, 0, 0]
        if extras:
            assert len(out[4]) == 2 and np.array_equal(out[4][0].numpy(), img) and np.array_equal(out[4][1].numpy(), xyz)
    # the results are COPIES, also for ONE frame (a one-row slice of the record is contiguous as it stands: numpy would hand
    # back a view of the pinned buffer, which the engine's next step overwrites)
    for fields in (1, 3):
        rec = np.zeros((2, record_bytes(fields)), dtype=np.uint8)
        rec[0, 40:292] = kp[0].reshape(63).view(np.uint8)
        rec[0, :32] = box[0].view(np.uint8)
        buf = torch.from_numpy(rec)
        out = read_host_record(buf, 1, extras=fields == 3)
        rec[:] = 0xFF
        assert np.array_equal(out[0].numpy()[0], kp[0]) and np.array_equal(out[2].numpy()[0], box[0])
        if fields == 3:
            assert not out[4][0].numpy().any() and not out[4][1].numpy().any()


def test_fragment_order_layout():
    """ops.fragment_order: the split filter bank [Fout, K/32, 2, 32] re-ordered so that an MFMA B fragment of 16 columns x 32
    channels is 64 lanes x 8 halves of contiguous memory -- element i of lane l = column 16 nt + (l & 15), channel 32 kt +
    8 (l >> 4) + i; columns behind Fout are zero (hn_graph_conv_cheby3_f16x3 with w_frag = 1 reads it without a bounds test)."""
    from hn_amd import ops
    g = torch.Generator().manual_seed(3)
    for fout, kt in ((3, 1), (40, 3), (256, 24)):
        w16 = torch.randn((fout, kt, 2, 32), generator=g).half()
        f = ops.fragment_order(w16)
        nt = (fout + 15) // 16
        assert tuple(f.shape) == (nt, kt, 2, 64, 8) and f.is_contiguous()
        for t in range(nt):
            for lane in (0, 1, 15, 16, 37, 63):
                col, chunk = t * 16 + (lane & 15), lane >> 4
                for k in range(kt):
                    for pl in (0, 1):
                        want = w16[col, k, pl, chunk * 8: chunk * 8 + 8] if col < fout else torch.zeros(8, dtype=torch.float16)
                        assert torch.equal(f[t, k, pl, lane], want)

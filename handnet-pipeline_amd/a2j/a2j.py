"""Drop-in `a2j.a2j.A2JModel` running on MI355X HIP kernels.

Same constructor, state_dict layout, call signature and return convention as the
reference class (a2j/a2j.py:212-250): `model(x[K,1,176,176]) -> FloatTensor[K,21,3]` on
the CPU (the reference ends with `.data.cpu()`, a2j/a2j.py:229).  Callers keep writing

    model = A2JModel(21, crop_height=176, crop_width=176, is_RGBD=False).cuda().eval()
    model.load_state_dict(torch.load(path, map_location="cpu")["model"], strict=False)
    jt_uvd = model(depth)[0]                                  # a2j_infer.py:25-28,58-60

`A2JModelLightning` (a2j/a2j.py:252-366) is provided for what the callers use it for -- `load_from_checkpoint(path)`
of a Lightning `.ckpt` (handnet_pipeline.py:28-29; commented alternative in a2j_infer.py:26, a2j_mesh.py:30) and
`forward` -- so that `from a2j.a2j import A2JModelLightning` (a2j_infer.py:12, a2j_mesh.py:15,
handnet_pipeline/handnet_pipeline.py:8) resolves -- and its evaluation hook `test_step` (a2j/a2j.py:333-359: forward ->
convert_joints on prediction and ground truth -> RMSE in mm -> the HPE evaluator's text file), the second caller SURVEY 8f #1
names, with the conversion in the aggregation's epilogue.  The training-side hooks raise.
Training (`gt is not None`) is out of scope and raises.
"""
from __future__ import annotations

import inspect
import os

import numpy as np
import torch

from hn_amd import synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd.pipeline import check_range_contract
from hn_amd.state import EngineOwner, build_state_tree


class A2JModel(EngineOwner):
    def __init__(self, num_classes, crop_height, crop_width, is_3D=True, is_RGBD=False, spatial_factor=0.5):
        super().__init__()
        self.is_3D = is_3D
        self.is_RGBD = is_RGBD
        self.num_classes = num_classes
        self.crop_height, self.crop_width = crop_height, crop_width
        # The reference starts from ImageNet weights fetched over the network
        # (a2j/a2j.py:188); offline, the state starts from the deterministic synthetic
        # checkpoint and is normally overwritten by load_state_dict().
        sd = synth.make_a2j_state_dict(seed=0, num_joints=num_classes, rgbd=is_RGBD)
        if not is_3D:  # a2j/a2j.py:219-220: no depth branch
            sd = {k: v for k, v in sd.items() if not k.startswith("DepthRegressionModel.")}
        tree = build_state_tree(sd)
        for name, child in tree.named_children():
            self.add_module(name, child)

    def engine(self) -> A2JEngine:
        if not self.is_3D:
            # Same failure as the reference: A2JModel builds post_process with its default is_3D=True
            # (a2j/a2j.py:223), so a two-head forward dies unpacking (cls, reg) into three names (a2j/anchor.py:59).
            raise ValueError("not enough values to unpack (expected 3, got 2) -- the reference's is_3D=False "
                             "forward fails the same way (a2j/a2j.py:223, a2j/anchor.py:58-59)")
        dev = self._require_gpu()
        if self._engine is None:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._engine = A2JEngine(sd, num_joints=self.num_classes, rgbd=self.is_RGBD, device=dev)
        return self._engine

    def forward_device(self, x, valid=None):
        """Same as forward() but leaves the [K,J,3] result on the GPU (no forced sync)."""
        return self.engine().forward(x, valid)

    def forward(self, x, gt=None):
        if gt is not None:
            raise NotImplementedError("training losses (a2j/anchor.py:84-152) are outside the inference hot path")
        # keypoints and the step's range-contract words in ONE device -> host copy (the reference's .data.cpu(), a2j.py:229);
        # an overflowing activation or a finite input beyond the fp16 range raises instead of returning inf / NaN or silently
        # wrong keypoints, a crop with NaN / inf pixels gives NaN keypoints like the reference (hn_amd.pipeline)
        kp, flags = self.engine().forward_flags(x)
        if flags is None:
            out = kp.cpu()
            check_range_contract(out, None, x)
            return out
        k, j3 = kp.shape[0], kp.shape[1] * kp.shape[2]
        flat = torch.cat([kp.contiguous().reshape(-1).view(torch.int32), flags[:3]]).cpu()
        out = flat[:k * j3].contiguous().view(torch.float32).reshape(kp.shape)
        check_range_contract(out, flat[k * j3:].tolist(), x)
        return out


    def forward_xyz(self, x, box, paras=None):
        """The forward of the callers that convert right away -- the evaluation loop (`test_step`, a2j/a2j.py:333-346) and the
        stand-alone demos (a2j_infer.py) -- with convert_joints + uvd2xyz in the aggregation's epilogue:
          x      [K,C,176,176] crops
          box    [K,4] the boxes the crops were cut with: float32 as the dataset hands them (fractional corners,
                 a2jdataset.py:293 -- every operation then stays in fp32, bit-identical to numpy on those operands) or int64 as
                 the detector path does (handnet_pipeline.py:88)
          paras  None, one camera (fx, fy, cx, cy), or [K,4] float32 per sample (a2jdataset.py:279)
        -> (crop uvd [K,J,3], image uvd [K,J,3], camera xyz in mm [K,J,3] or None): CPU tensors from ONE device -> host copy,
        under the same range contract as forward()."""
        eng = self.engine()
        dev = eng.device
        x = x.to(dev)
        k = x.shape[0]
        box = torch.as_tensor(box)
        conv = {}
        if box.dtype == torch.int64:
            conv["crop_box"] = box.reshape(k, 4).to(dev).contiguous()
        else:
            conv["sample_box"] = box.reshape(k, 4).to(torch.float32).to(dev).contiguous()
        if paras is not None:
            p = torch.as_tensor(paras, dtype=torch.float32)
            if p.numel() == 4 and (k != 1 or p.dim() == 1):
                conv["paras"] = [float(v) for v in p.reshape(4)]
            else:
                conv["sample_paras"] = p.reshape(k, 4).to(dev).contiguous()
        (kp, img, xyz), flags = eng.forward_flags(x, convert=conv)
        parts = [t.contiguous().reshape(-1).view(torch.int32) for t in (kp, img) + ((xyz,) if xyz is not None else ())]
        if flags is not None:
            parts.append(flags[:3])
        flat = torch.cat(parts).cpu()
        n = kp.numel()
        outs = [flat[i * n:(i + 1) * n].contiguous().view(torch.float32).reshape(kp.shape) for i in range(len(parts) - (flags is not None))]
        check_range_contract(outs[0], flat[-3:].tolist() if flags is not None else None, x)
        return outs[0], outs[1], (outs[2] if xyz is not None else None)


    def mesh(self, lifter, clamp: bool = True, perm_reverse=None):
        """The stand-alone mesh demo's loop body as ONE step (hn_amd.live.CropMeshEngine; a2j_mesh.py:58-80): this network, np.clip
        + convert_joints in the aggregation's epilogue, the lifter's input, Pose2Mesh, the caller's last lines, one copy.
        lifter: the drop-in `models.pose2mesh_net.get_model(...)` module (on the GPU) or a Pose2MeshEngine."""
        from hn_amd.live import CropMeshEngine
        return CropMeshEngine(self.engine(), lifter.engine() if hasattr(lifter, "engine") else lifter, clamp, perm_reverse)


class A2JModelLightning(EngineOwner):
    """Inference-side stand-in for the reference's LightningModule (a2j/a2j.py:252-366): same constructor
    arguments, `.a2j` = the HIP-backed A2JModel, state_dict keys `a2j.*` (the layout of a Lightning checkpoint's
    `state_dict`), `load_from_checkpoint`, `forward`, and the evaluation hooks `test_step` / `test_epoch_end` / `log`.
    pytorch-lightning is not needed (and not installed here); the training-side hooks raise."""

    def __init__(self, num_classes: int = 21, crop_height: int = 176, crop_width: int = 176, is_3D: bool = True,
                 is_RGBD: bool = False, spatial_factor: float = 0.5, display_freq: int = 5000,
                 output_dir: str = "models/a2j"):
        super().__init__()
        self.hparams = dict(num_classes=num_classes, crop_height=crop_height, crop_width=crop_width, is_3D=is_3D,
                            is_RGBD=is_RGBD, spatial_factor=spatial_factor, display_freq=display_freq,
                            output_dir=output_dir)
        self.a2j = A2JModel(num_classes, crop_height, crop_width, is_3D, is_RGBD, spatial_factor)
        # the reference ctor unconditionally loads this file (a2j/a2j.py:278); offline it is absent, and
        # load_from_checkpoint overwrites the weights anyway
        pre = "models/a2j_dexycb_1/a2j_25.pth"
        if os.path.exists(pre):
            self.a2j.load_state_dict(torch.load(pre, map_location="cpu")["model"], strict=False)
        self.rgbd = is_RGBD
        self.display_freq = display_freq
        self.output_dir = output_dir

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict: bool = True, **kwargs):
        """LightningModule.load_from_checkpoint: constructor arguments come from the file's `hyper_parameters`
        (save_hyperparameters, a2j/a2j.py:276) overridden by kwargs; weights from `state_dict` (keys `a2j.*`)."""
        ckpt = torch.load(checkpoint_path, map_location="cpu")
        if "state_dict" not in ckpt:
            raise KeyError(f"{checkpoint_path} is not a Lightning checkpoint (no 'state_dict'); plain A2J files "
                           "are loaded with A2JModel.load_state_dict(ckpt['model'])")
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(kwargs)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in hp.items() if k in accepted})
        sd = ckpt["state_dict"]
        own = set(model.state_dict().keys())
        # strict (Lightning's default) = every weight of THIS module must be in the file; entries this inference-only
        # module does not carry (the anchor / threshold buffers of A2J_loss and post_process, a2j/anchor.py:48-49,88-90,
        # optimizer-side state) are ignored
        missing = sorted(k for k in own if k not in sd and not k.endswith("num_batches_tracked"))
        if strict and missing:
            raise KeyError(f"{checkpoint_path}: state_dict lacks {len(missing)} entries, e.g. {missing[:3]}")
        model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        if map_location is not None and str(map_location) != "cpu":
            model = model.to(map_location)
        return model

    def engine(self):
        return self.a2j.engine()

    def forward(self, x, gt=None):
        return self.a2j(x, gt)

    def forward_device(self, x, valid=None):
        return self.a2j.forward_device(x, valid)

    current_epoch = 0      # (Lightning's trainer sets it; `test_step` names its output file after it, a2j/a2j.py:356)

    def log(self, name, value):
        """Lightning's self.log, reduced to what the evaluation loop needs: values per name, in call order (`logged`)."""
        self.__dict__.setdefault("logged", {}).setdefault(name, []).append(float(value))

    def test_step(self, batch, batch_idx):
        """a2j/a2j.py:333-359, the evaluation caller of the path: forward -> convert_joints (prediction AND ground truth, the
        dataset's float32 box and the sample's intrinsics) -> RMSE in mm -> one text line per sample for the HPE evaluator.
        The prediction's conversion rides in the aggregation's epilogue (it comes back in the forward's one copy), the ground
        truth's is one small launch; the RMSE and the text are numpy on those values exactly as the reference writes them.
        The reference's convert_joints reshapes the whole batch against ONE box (batch size 1 only); here every sample is
        converted with its own box and intrinsics and gets its own line."""
        from hn_amd import ops
        im, jt_uvd_gt, dexycb_id, _color_im, box, paras, combined_im = batch
        x = combined_im if self.rgbd else im
        k = x.shape[0]
        _kp, _img, pred_xyz = self.a2j.forward_xyz(x, torch.as_tensor(box, dtype=torch.float32), torch.as_tensor(paras, dtype=torch.float32).reshape(k, 4))
        dev = self.a2j.engine().device
        gt = torch.as_tensor(jt_uvd_gt, dtype=torch.float32).reshape(k, -1, 3).to(dev).contiguous()
        _, gt_xyz = ops.convert_joints_samples(gt, torch.as_tensor(box, dtype=torch.float32).reshape(k, 4).to(dev).contiguous(),
                                               torch.as_tensor(paras, dtype=torch.float32).reshape(k, 4).to(dev).contiguous(),
                                               want_image=False)
        jt_xyz_pred, jt_xyz_gt = pred_xyz.numpy().reshape(-1, 3), gt_xyz.cpu().numpy().reshape(-1, 3)
        rmse = np.sqrt(np.mean(np.square(jt_xyz_gt - jt_xyz_pred)))
        self.log("test_rmse", rmse)
        os.makedirs(os.path.join(self.output_dir, "a2j_test_metrics"), exist_ok=True)
        epoch_output = os.path.join(self.output_dir, f"a2j_test_metrics/s0_test_{self.current_epoch}.txt")
        j = jt_xyz_pred.shape[0] // k
        with open(epoch_output, "a") as output:
            for i in range(k):
                # "x,y,z,x,y,z,..." in the digits numpy 1.x prints for a float32 inside a list (= str() of the scalar: shortest
                # round trip), no spaces, no trailing comma (a2j/a2j.py:351-354)
                j_text = ",".join(str(v) for v in jt_xyz_pred[i * j:(i + 1) * j].reshape(-1))
                ident = dexycb_id[i]
                ident = ident.cpu().numpy() if torch.is_tensor(ident) else np.asarray(ident)
                print(str(ident)[1:-1] + "," + j_text, file=output)
        return rmse

    def test_epoch_end(self, outputs):
        """a2j/a2j.py:361-363: hands the epoch's file to dex_ycb_toolkit's HPEEvaluator (not part of this image: raises a clear
        ImportError when the toolkit is absent; the file `test_step` wrote is complete either way)."""
        try:
            from dex_ycb_toolkit.hpe_eval import HPEEvaluator
        except ImportError as e:
            raise ImportError("test_epoch_end needs dex_ycb_toolkit (HPEEvaluator); the predictions are in "
                              f"{os.path.join(self.output_dir, 'a2j_test_metrics')}") from e
        hpe_eval = HPEEvaluator("s0_test")
        hpe_eval.evaluate(self.current_epoch, os.path.join(self.output_dir, f"a2j_test_metrics/s0_test_{self.current_epoch}.txt"),
                          os.path.join(self.output_dir, "dexycb_metrics/"))

    def _no_training(self, *a, **k):
        raise NotImplementedError("the training-side hooks of A2JModelLightning (training_step / validation_step need the A2J "
                                  "loss, a2j/a2j.py:283-331; configure_optimizers) are outside the inference hot path")

    training_step = validation_step = configure_optimizers = _no_training


def convert_joints(jt_uvd_pred, jt_uvd_gt, box, paras, cropWidth, cropHeight):
    """crop-(u,v,d) -> image (u,v,d) -> camera xyz in mm; same contract as a2j/a2j.py:17-43."""
    def one(jt):
        jt = np.asarray(jt).reshape(-1, 3)
        b = np.asarray(box).reshape(4)
        out = np.ones_like(jt)
        out[:, 0] = jt[:, 0] * (b[2] - b[0]) / cropWidth + b[0]
        out[:, 1] = jt[:, 1] * (b[3] - b[1]) / cropHeight + b[1]
        out[:, 2] = jt[:, 2]
        if paras is not None:
            p = np.asarray(paras).reshape(4)
            out[:, :2] = (out[:, :2] - p[2:]) * out[:, 2:] / p[:2]
            out = out.astype(np.float32) * 1000.0
        return out

    pred = one(jt_uvd_pred)
    if jt_uvd_gt is not None:
        return pred, one(jt_uvd_gt)
    return pred

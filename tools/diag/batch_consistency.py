"""Frame 0 must not depend on the batch it travels in beyond fp32 rounding: runs the detector at several batch sizes and reports,
for frame 0, the largest score difference against the batch-1 run and the crop box (development aid; forms via HN_* variables)."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
from hn_amd import forms, synth  # noqa: E402
from hn_amd.a2j_engine import A2JEngine  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402
from hn_amd.pipeline import HandNetEngine  # noqa: E402

print("forms:", forms.apply_env())
fe = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, device="cuda")
eng = HandNetEngine(fe, A2JEngine(synth.make_a2j_state_dict(0), device="cuda"), 3)
rgb, depth = synth.make_rgb(32, seed=1000).cuda(), synth.make_depth(32, seed=2000).cuda()


def scores(b):
    cls_lr, reg_ctr, _, _ = fe.forward_heads(rgb[:b])
    cls = torch.cat([t[0].reshape(-1, t.shape[-1])[:, :3] for t in cls_lr])
    ctr = torch.cat([t[0].reshape(-1, 5)[:, 4:5] for t in reg_ctr])
    return torch.sqrt(torch.sigmoid(cls) * torch.sigmoid(ctr)).max(dim=-1)[0]


base = scores(1)
for b in (1, 2, 3, 4, 8, 12, 16, 24, 32):
    out = eng.forward_device(rgb[:b], depth[:b])
    s = scores(b)
    print(f"batch {b:2d}: frame-0 crop {out.crop_box[0].tolist()}  max |score - batch-1 score| {float((s - base).abs().max()):.2e}  "
          f"kp diff {float((out.keypoints[0] - eng.forward_device(rgb[:1], depth[:1]).keypoints[0]).abs().max()):.2e}", flush=True)

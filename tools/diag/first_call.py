"""Is the FIRST forward of a fresh process (with another process on the card) different from the later ones?  (round 4:
tests/test_dist_gpu.py two-rank case failed intermittently on frame 0 of a rank's single forward.)  usage: first_call.py <tag>"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
from hn_amd import forms, ops, synth  # noqa: E402
from hn_amd.a2j_engine import A2JEngine  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402
from hn_amd.pipeline import HandNetEngine  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "p"
print(tag, "forms:", forms.apply_env(), flush=True)
eng = HandNetEngine(FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, device="cuda"), A2JEngine(synth.make_a2j_state_dict(0), device="cuda"), 3)
rgb, depth = synth.make_rgb(16, seed=1000).cuda(), synth.make_depth(16, seed=2000).cuda()
outs = []
for it in range(4):
    o = eng.forward_device(rgb, depth)
    torch.cuda.synchronize()
    outs.append((o.crop_box.clone(), o.keypoints.clone(), o.candidates.count.clone(), o.detections.count.clone(),
                 o.detections.scores[:, :8].clone()))
for it in range(1, 4):
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[it]))
    print(tag, f"call 0 vs call {it}: {'identical' if same else 'DIFFERENT'}", flush=True)
    if not same:
        for name, a, b in zip(("crop_box", "keypoints", "cand count", "det count", "top scores"), outs[0], outs[it]):
            if not torch.equal(a, b):
                rows = torch.nonzero((a != b).reshape(a.shape[0], -1).any(dim=1)).flatten().tolist()
                print(tag, "   ", name, "differs in frames", rows, flush=True)
print(tag, "frame 0 crop (call 0):", outs[0][0][0].tolist(), flush=True)

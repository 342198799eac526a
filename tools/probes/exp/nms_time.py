"""Time hn_fcos_nms alone over candidate counts (where do the 65 us of the batch-1 frame's NMS go: sort or greedy tiles?)."""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

cap = 17850
g = torch.Generator().manual_seed(3)
for k in (32, 64, 65, 128, 192, 260, 320, 512, 513, 1024, 2048, 2049, 4096):
    ctr = torch.rand((k, 2), generator=g) * 600.0
    wh = 20.0 + torch.rand((k, 2), generator=g) * 80.0
    cand = ops.alloc_candidates(1, cap, "cuda")
    cand.boxes[0, :k] = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).cuda()
    cand.scores[0, :k] = (0.7 + 0.3 * torch.rand((k,), generator=g)).cuda()
    cand.labels[0, :k] = torch.randint(0, 3, (k,), generator=g).int().cuda()
    cand.sides[0, :k] = 0
    cand.level[0, :k] = 0
    cand.count[0] = k
    det = ops.fcos_nms(cand, 0.3, 0.8, 0.8)
    scratch = torch.empty((ops._lib.load().hn_fcos_nms_scratch_bytes(1, cap),), device="cuda", dtype=torch.uint8)
    for _ in range(5):
        ops.fcos_nms(cand, 0.3, 0.8, 0.8, scratch=scratch, out=det)
    t = ops.HipTimer()
    reps = 100
    t.start()
    for _ in range(reps):
        ops.fcos_nms(cand, 0.3, 0.8, 0.8, scratch=scratch, out=det)
    t.stop()
    torch.cuda.synchronize()
    print(f"K {k:5d}: {t.elapsed_ms() / reps * 1e3:8.1f} us, kept {int(det.count[0])}", flush=True)

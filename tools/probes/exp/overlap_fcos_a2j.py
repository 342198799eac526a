"""Experiment: does the A2J half of step i hide under the FCOS half of step i+1 when the two run on two streams?
(A2J at batch 32 is 2.1 ms at 0.06-0.11 of the MFMA peak: launch- and latency-bound; the FCOS towers are power-bound.)
usage: overlap_fcos_a2j.py [batch] [steps]"""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
from hn_amd import ops, synth  # noqa: E402
from hn_amd.a2j_engine import A2JEngine  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402
from hn_amd.pipeline import CROP, HandNetEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fcos = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, device="cuda")
a2j = A2JEngine(synth.make_a2j_state_dict(0), device="cuda")
eng = HandNetEngine(fcos, a2j, 3)
eng.note_range = False
rgb = synth.make_rgb(n, seed=1000).cuda()
depth = synth.make_depth(n, seed=2000).cuda()
ops.range_check_enable(False)


def serial(k):
    outs = []
    for _ in range(k):
        det, cand = fcos.detect(rgb)
        box, has, crops = ops.crop_resize(det, 2, depth, CROP, 4, reorder_bgr=False)
        outs.append(a2j.forward_nhwc(crops, valid=has))
    return outs


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def piped(k):
    outs, keep = [], []
    for _ in range(k):
        with torch.cuda.stream(sa):
            det, cand = fcos.detect(rgb)
            box, has, crops = ops.crop_resize(det, 2, depth, CROP, 4, reorder_bgr=False)
            ev = torch.cuda.Event()
            ev.record(sa)
        keep.append((det, cand, box, has, crops))
        with torch.cuda.stream(sb):
            sb.wait_event(ev)
            outs.append(a2j.forward_nhwc(crops, valid=has))
    return outs, keep


def timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, r


serial(3)
piped(3)
torch.cuda.synchronize()
for rep in range(3):
    ts, rs = timed(serial, steps)
    tp, (rp, _) = timed(piped, steps)
    same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(rs, rp))
    print(f"batch {n}: serial {ts:.3f} ms/step ({n / ts * 1e3:.1f} frames/s)   two streams {tp:.3f} ms/step ({n / tp * 1e3:.1f} frames/s)"
          f"   keypoints bit-identical: {same}", flush=True)

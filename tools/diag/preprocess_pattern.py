"""Characterise the run-to-run differences of the tiled preprocess kernel (tools/diag/preprocess_repeat.py found them): which
(plane, channel, lane, wave, k) positions differ from the per-pixel kernel, and what the wrong value is (a neighbour's value?)."""
import sys
from collections import Counter
sys.path.insert(0, "handnet-pipeline_amd")
import torch
from hn_amd import ops, synth
from hn_amd.fcos_engine import IMAGE_MEAN, IMAGE_STD

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rgb = synth.make_rgb(n, seed=1000).cuda()
args = (rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
ops.set_form("preprocess_generic", True)
ref = ops.fcos_preprocess_split(*args)
ref2 = ops.fcos_preprocess_split(*args)
torch.cuda.synchronize()
print("per-pixel kernel repeatable:", torch.equal(ref, ref2), flush=True)
ops.set_form("preprocess_generic", False)
hist = Counter()
tot = 0
for it in range(passes):
    t = ops.fcos_preprocess_split(*args)
    torch.cuda.synchronize()
    idx = torch.nonzero(t != ref).cpu()
    tot += len(idx)
    print("pass", it, "halves differing from the per-pixel kernel:", len(idx), flush=True)
    for p, im, r, c, ch in idx.tolist():
        hist[("plane", p)] += 1
        hist[("ch", ch)] += 1
        hist[("j", c & 1)] += 1
        hist[("lane/16", ((c % 128) // 2) // 16)] += 1
        hist[("wave", (r % 8) % 4)] += 1
        hist[("k", (r % 8) // 4)] += 1
    if it == 0 and len(idx):
        tc, rc = t.cpu().float(), ref.cpu().float()
        shown = 0
        for p, im, r, c, ch in idx.tolist()[:4000:400]:
            got, want = float(tc[p, im, r, c, ch]), float(rc[p, im, r, c, ch])
            where = []
            for dr in range(-8, 9):
                for dc in range(-4, 5):
                    for dch in range(3):
                        rr, cc = r + dr, c + dc
                        if 0 <= rr < rc.shape[2] and 0 <= cc < rc.shape[3] and float(rc[p, im, rr, cc, dch]) == got and (dr, dc, dch) != (0, 0, ch):
                            where.append((dr, dc, dch))
            print("  at", (p, im, r, c, ch), "lane", (c % 128) // 2, "got", got, "want", want, "got == ref at offsets (dr, dc, ch):", where[:6], flush=True)
print("total", tot)
for k in sorted(hist, key=str):
    print(k, hist[k])

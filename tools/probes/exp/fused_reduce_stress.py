"""Stress of the split-K reduction inside the last workgroup (per-wave tickets, sc1 partial planes): many repeats of split
launches of every fused tile form, every result compared bit for bit with the separate-reduction form.  Prints one line per
case: launches, mismatching launches.   usage (GPU box): python tools/probes/exp/fused_reduce_stress.py [reps]"""
import sys
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
from hn_amd import ops  # noqa: E402
from hn_amd.weights import split_f16x3  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CASES = [((1, 25, 34, 512, 512, 3, 1, 1, 1), 7, 16), ((1, 25, 34, 512, 512, 3, 1, 1, 1), 7, 3), ((1, 11, 11, 2048, 512, 3, 1, 1, 1), 7, 8),
         ((1, 50, 68, 256, 256, 3, 1, 1, 1), 3, 4), ((1, 50, 68, 256, 256, 3, 1, 1, 1), 12, 3), ((1, 100, 136, 128, 128, 3, 1, 1, 1), 6, 2),
         ((4, 50, 68, 256, 256, 3, 1, 1, 1), 3, 2), ((32, 11, 11, 512, 512, 3, 1, 1, 1), 6, 12), ((64, 11, 11, 256, 256, 3, 1, 1, 1), 6, 4)]
g = torch.Generator().manual_seed(7)
bad_total = 0
for (n, h, w, cin, cout, r, stride, pad, dil), tile, splits in CASES:
    x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
    wt = torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5
    b = torch.randn((cout,), generator=g).cuda()
    oh, ow = ops.conv_out_size(h, w, r, r, stride, pad, dil)
    res = ops.to_split(torch.randn((n, oh, ow, cout), generator=g).cuda())
    kw = dict(stride=stride, pad=pad, dil=dil, relu=True, tile=tile, w16=split_f16x3(wt).cuda(), force_splits=splits, residual=res,
              out_split=True)
    ops.set_form("conv_no_fused_reduce", True)
    ref = ops.conv2d_nhwc(x, wt.cuda(), b, **kw).clone()
    ops.set_form("conv_no_fused_reduce", False)
    bad = 0
    outs = []
    for i in range(reps):
        outs.append(ops.conv2d_nhwc(x, wt.cuda(), b, **kw))
        if len(outs) == 20:   # compare in batches: the launches themselves run back to back
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
            outs = []
    bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    bad_total += bad
    print(f"{(n, h, w, cin, cout, r)} tile {tile} splits {splits}: {reps} launches, {bad} mismatching", flush=True)
print("TOTAL mismatching launches:", bad_total)
sys.exit(1 if bad_total else 0)

"""Stress the heterogeneous launch against separate launches while another process keeps the GPU busy (an intermittent
difference showed up only with two processes on the card: tests/test_dist_gpu.py, round 4).
usage: multi_stress.py [iters]     (run a second copy with the argument `load` to create the contention)"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
from hn_amd import forms, ops  # noqa: E402
from hn_amd.weights import ConvW  # noqa: E402

forms.apply_env()


def member(seed, n, h, w, cin, cout, r, stride=1, relu=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, h, w, cin), generator=g).cuda()
    wt = torch.randn((cout, r, r, cin), generator=g) * (2.0 / (r * r * cin)) ** 0.5
    cw = ConvW(wt, torch.randn((cout,), generator=g) * 0.1, stride, r // 2, 1).to("cuda")
    return ops.to_split(x), cw, dict(relu=relu)


if len(sys.argv) > 1 and sys.argv[1] == "load":
    x, cw, _ = member(99, 8, 100, 136, 256, 256, 3)
    for _ in range(4000):
        ops.conv2d_nhwc(x, cw.w, cw.bias, pad=1, w16=cw.w16, out_split=True)
    torch.cuda.synchronize()
    sys.exit(0)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 16
cases = {
    "layer2.0": lambda: [member(1, n, 200, 272, 64, 128, 3, 2), member(2, n, 200, 272, 64, 128, 1, 2, relu=False)],
    "layer3.0": lambda: [member(3, n, 100, 136, 128, 256, 3, 2), member(4, n, 100, 136, 128, 256, 1, 2, relu=False)],
    "layer4.0": lambda: [member(5, n, 50, 68, 256, 512, 3, 2), member(6, n, 50, 68, 256, 512, 1, 2, relu=False)],
}
for name, make in cases.items():
    items = make()
    # both members read the SAME input in the engines
    items[1] = (items[0][0], items[1][1], items[1][2])
    ref = [ops.conv2d_nhwc(x, cw.w, cw.bias, stride=cw.stride, pad=cw.pad, w16=cw.w16, relu=o["relu"], out_split=True)
           for x, cw, o in items]
    bad = [0, 0]
    first = None
    for it in range(iters):
        got = ops.conv2d_nhwc_multi(items)
        for m in range(2):
            if not torch.equal(got[m], ref[m]):
                bad[m] += 1
                if first is None:
                    d = (ops.from_split(got[m]) - ops.from_split(ref[m])).abs()
                    idx = torch.nonzero(d > 0)
                    first = (it, m, int(idx.shape[0]), idx[:3].tolist(), float(d.max()))
    print(f"{name}: {iters} iterations, differing outputs per member {bad}, first {first}", flush=True)

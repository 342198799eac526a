"""A/B of the 128x256 eight-wave tile (HN_TILE_128x256_W8: the A operand of a Cout = 256 layer fetched once) against the
128x128 and 256x128w8 tiles on the tower / layer3 / layer4 shapes; outputs must be bit-identical (same k order per output)."""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops
from hn_amd.weights import split_f16x3

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [(32, 100, 136, 256, 256, 3), (32, 50, 68, 256, 256, 3), (32, 50, 68, 256, 512, 3), (32, 25, 34, 512, 512, 3),
          (32, 100, 136, 128, 256, 1), (4, 50, 68, 256, 256, 3)]
TILES = [1, 9, 11]
if len(sys.argv) > 2 and sys.argv[2] == "tower":   # (the counter passes: one shape, two tiles)
    SHAPES, TILES = SHAPES[:1], [1, 11]
print("us per launch by tile", [ops.TILE_NAMES[t] for t in TILES], flush=True)
for n, h, w, cin, cout, r in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
    wt = torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5
    w16 = split_f16x3(wt).cuda()
    wt = wt.cuda()
    b = torch.randn((cout,), generator=g).cuda()
    res, outs = {}, {}
    for rep in range(2):
        for t in TILES:
            kw = dict(stride=1, pad=r // 2, dil=1, relu=True, tile=t, w16=w16, out_split=True)
            y = ops.conv2d_nhwc(x, wt, b, **kw)
            outs[t] = y.clone()
            kw["out"] = y
            for _ in range(3):
                ops.conv2d_nhwc(x, wt, b, **kw)
            tm = ops.HipTimer()
            tm.start()
            for _ in range(iters):
                ops.conv2d_nhwc(x, wt, b, **kw)
            tm.stop()
            res[t] = min(res.get(t, 1e30), tm.elapsed_ms() * 1e3 / iters)
    gf = 2.0 * n * h * w * cin * cout * r * r / 1e9
    same = all(torch.equal(outs[1], outs[t]) for t in TILES)
    print(f"{(n, h, w, cin, cout, r)}: " + "  ".join(f"{ops.TILE_NAMES[t]} {res[t]:8.1f} us ({gf / res[t] * 1e3:6.1f} TFLOP/s)" for t in TILES)
          + f"   bit-identical: {same}", flush=True)

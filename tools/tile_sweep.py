"""Time every f16x3 tile on every distinct conv shape of the pipeline (development aid for pick_tile).
usage: tile_sweep.py [batch] [iters]  -> table on stdout (best tile per shape vs what the heuristic picks)"""
import sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops, synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd.fcos_engine import FCOSEngine
from hn_amd.pipeline import HandNetEngine
from hn_amd.weights import split_f16x3
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
only = sys.argv[3] if len(sys.argv) > 3 else "all"
fcos = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3)
a2j = A2JEngine(synth.make_a2j_state_dict(0))
eng = HandNetEngine(fcos, a2j, 3)
rgb = synth.make_rgb(batch, seed=1000).cuda()
depth = synth.make_depth(batch, seed=2000).cuda()
eng.forward_device(rgb, depth)
torch.cuda.synchronize()
ops.CONV_PROFILE = []
eng.forward_device(rgb, depth)
torch.cuda.synchronize()
recs, ops.CONV_PROFILE = ops.CONV_PROFILE, None
shapes = {}
for kind, macs, timer, shape, _stage in recs:
    if kind[0] == "f16x3" and shape[3] % 32 == 0:
        if isinstance(kind[1], int):   # (thin-N / direct kernels record a name: not igemm tiles)
            shapes.setdefault(shape, [kind[1] & 0xFF, 0])[1] += 1
del eng, fcos, a2j
torch.cuda.empty_cache()
TILES = [1, 2, 3, 4, 6, 7, 8, 12]
# small kernels in isolation leave the GPU in a low DPM state and time 2-3x slow: keep the clocks up with a
# heavy convolution right before every timed loop
_hx = ops.to_split(torch.randn((8, 100, 136, 256), generator=torch.Generator().manual_seed(1)).cuda())
_hw = torch.randn((256, 3, 3, 256), generator=torch.Generator().manual_seed(2)) * 0.02
_hw16 = split_f16x3(_hw).cuda()
_hw = _hw.cuda()


def heat():
    for _ in range(12):
        ops.conv2d_nhwc(_hx, _hw, None, pad=1, w16=_hw16, out_split=True)

print(f"# batch {batch}: {len(shapes)} distinct shapes; us per launch by tile {[ops.TILE_NAMES[t] for t in TILES]}")
tot_now = tot_best = 0.0
for shape, (picked, calls) in sorted(shapes.items(), key=lambda kv: kv[0]):
    n, h, w, cin, cout, r, stride, dil = shape
    if only == "small" and n * h * w > 40000:
        continue
    pad = dil * (r // 2)
    g = torch.Generator().manual_seed(0)
    x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
    wt = (torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5)
    w16 = split_f16x3(wt).cuda()
    wt = wt.cuda()
    b = torch.randn((cout,), generator=g).cuda()
    osplit = cout % 32 == 0
    res = {}
    for t in TILES:
        if (t == 8 and cout > 64) or (t == 4 and cout > 32):
            continue
        kw = dict(stride=stride, pad=pad, dil=dil, relu=True, tile=t, w16=w16, out_split=osplit)
        y = ops.conv2d_nhwc(x, wt, b, **kw)
        kw["out"] = y
        for _ in range(3):
            ops.conv2d_nhwc(x, wt, b, **kw)
        heat()
        tm = ops.HipTimer()
        tm.start()
        for _ in range(iters):
            ops.conv2d_nhwc(x, wt, b, **kw)
        tm.stop()
        res[t] = tm.elapsed_ms() * 1e3 / iters
    best = min(res, key=res.get)
    tot_now += calls * res.get(picked, res[best])
    tot_best += calls * res[best]
    row = " ".join(f"{res.get(t, float('nan')):8.1f}" for t in TILES)
    flag = "" if best == picked or res[picked] <= 1.03 * res[best] else f"  <-- best {ops.TILE_NAMES[best]} ({res[picked] / res[best]:.2f}x)"
    print(f"{str(shape):44s} x{calls:2d} picked {ops.TILE_NAMES[picked]:8s} | {row}{flag}")
print(f"# sum over calls: picked {tot_now / 1e3:.3f} ms, best-per-shape {tot_best / 1e3:.3f} ms")

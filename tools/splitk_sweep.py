"""Time every split-K factor on every distinct small-grid conv shape of the pipeline (development aid for plan_splits).
usage: splitk_sweep.py [batch] [iters]  -> table on stdout: us per conv (+ reduction) for 1..16 splits, the plan's own choice"""
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

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fcos = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, forms=dict(conv_multi_fcos=False))
a2j = A2JEngine(synth.make_a2j_state_dict(0), forms=dict(conv_multi=False))
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
for kind, macs, timer, shape, stage in recs:
    if kind[0] == "f16x3" and shape[3] % 32 == 0 and shape[0] * shape[1] * shape[2] < 40000 and shape[4] % 32 == 0:
        shapes.setdefault(shape, [kind[1], 0])[1] += 1
del eng, fcos, a2j
torch.cuda.empty_cache()
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]
_hx = ops.to_split(torch.randn((8, 100, 136, 256), generator=torch.Generator().manual_seed(1)).cuda())
_hw = torch.randn((256, 3, 3, 256), generator=torch.Generator().manual_seed(2)) * 0.02
_hw16 = split_f16x3(_hw).cuda()
_hw = _hw.cuda()


def heat():
    for _ in range(12):
        ops.conv2d_nhwc(_hx, _hw, None, pad=1, w16=_hw16, out_split=True)


def timed(fn):
    for _ in range(3):
        fn()
    heat()
    tm = ops.HipTimer()
    tm.start()
    for _ in range(iters):
        fn()
    tm.stop()
    return tm.elapsed_ms() * 1e3 / iters


print(f"# batch {batch}: {len(shapes)} small-grid shapes; us per conv (+ reduction) by split count {SPLITS}, then the plan's own")
tot_plan = tot_best = 0.0
for shape, (picked, calls) in sorted(shapes.items(), key=lambda kv: kv[0]):
    n, h, w, cin, cout, r, stride, dil = shape
    pad = dil * (r // 2)
    g = torch.Generator().manual_seed(0)
    x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
    wt = (torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5)
    w16 = split_f16x3(wt).cuda()
    wt = wt.cuda()
    b = torch.randn((cout,), generator=g).cuda()
    kw = dict(stride=stride, pad=pad, dil=dil, relu=True, w16=w16, out_split=True)
    y = ops.conv2d_nhwc(x, wt, b, **kw)
    kw["out"] = y
    res = {}
    for s in SPLITS:
        if s > r * r * cin // 32:
            continue
        res[s] = timed(lambda: ops.conv2d_nhwc(x, wt, b, force_splits=s, **kw))
    plan = timed(lambda: ops.conv2d_nhwc(x, wt, b, **kw))
    best = min(res, key=res.get)
    tot_plan += calls * plan
    tot_best += calls * res[best]
    row = " ".join(f"{res.get(s, float('nan')):7.1f}" for s in SPLITS)
    print(f"{str(shape):44s} x{calls:2d} tile {ops.tile_name(picked):9s} {row} | plan {plan:7.1f}  best s={best:2d} {res[best]:7.1f}"
          + ("" if plan <= 1.05 * res[best] else f"  <-- {plan / res[best]:.2f}x"), flush=True)
print(f"# sum over the step: plan {tot_plan:.0f} us, best {tot_best:.0f} us")

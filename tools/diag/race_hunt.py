"""Hunt a nondeterminism that only shows with a second process on the card: `race_hunt.py load <seconds>` keeps the GPU busy;
`race_hunt.py check <iters>` runs the FCOS stages again and again on fixed inputs and reports the FIRST stage whose output is not
bit-identical to the first pass (kernel forms through the HN_* variables, hn_amd/forms.py)."""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
from hn_amd import forms, ops, synth  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402
from hn_amd.weights import ConvW  # noqa: E402

mode = sys.argv[1]
if mode == "load":
    g = torch.Generator().manual_seed(1)
    x = ops.to_split(torch.randn((8, 100, 136, 256), generator=g).cuda())
    wt = torch.randn((256, 3, 3, 256), generator=g) * 0.02
    cw = ConvW(wt, None, 1, 1, 1).to("cuda")
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(50):
            ops.conv2d_nhwc(x, cw.w, None, pad=1, w16=cw.w16, out_split=True)
        torch.cuda.synchronize()
    sys.exit(0)

print("forms:", forms.apply_env(), flush=True)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
eng = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, device="cuda")
rgb = synth.make_rgb(n, seed=1000).cuda()
oh, ow, ph, pw = eng.geometry(480, 640)


def stages():
    out = []
    with ops.f16_terms(3):
        x16 = ops.fcos_preprocess_split(rgb, oh, ow, ph, pw, eng.image_mean, eng.image_std)
        out.append(("preprocess (interior)", x16[:, :, 3:-3, 3:-3, :3].clone()))   # (padding lanes of the buffer are never written)
        x = ops.conv_stem_pool_split(x16, eng.stem16.w16, eng.stem16.bias, 64, r=7, stride=2)
        out.append(("stem+pool", x))
        feats = []
        for bi, blk in enumerate(eng.blocks):
            if blk["ds"] is not None and eng.multi:
                o, idn = ops.conv2d_nhwc_multi([(x, blk["c1"], dict(relu=True)), (x, blk["ds"], dict(relu=False))])
                out.append((f"block {bi} c1 (multi)", o))
                out.append((f"block {bi} ds (multi)", idn))
            else:
                o = eng._conv(x, blk["c1"], relu=True)
                out.append((f"block {bi} c1", o))
                idn = eng._conv(x, blk["ds"]) if blk["ds"] is not None else x
                if blk["ds"] is not None:
                    out.append((f"block {bi} ds", idn))
            x = eng._conv(o, blk["c2"], relu=True, residual=idn)
            out.append((f"block {bi} c2", x))
            if blk["last"] and blk["layer"] >= 2:
                feats.append(x)
        c3, c4, c5 = feats
        lat5 = eng._conv(c5, eng.inner[2])
        lat4 = eng._conv(c4, eng.inner[1], residual=lat5, res_upsample=True)
        lat3 = eng._conv(c3, eng.inner[0], residual=lat4, res_upsample=True)
        out += [("lat5", lat5), ("lat4", lat4), ("lat3", lat3)]
        fp = ops.conv2d_nhwc_grouped([lat3, lat4, lat5], eng.layer, pad=1, out_split=True)
        out += [(f"fpn out {i}", t) for i, t in enumerate(fp)]
        heads = eng.heads(fp)
        for i, (c, r, _) in enumerate(heads):
            out += [(f"cls_lr {i}", c), (f"reg_ctr {i}", r)]
    torch.cuda.synchronize()
    return out


ref = stages()
bad = {}
for it in range(iters):
    cur = stages()
    for (name, a), (_, b) in zip(ref, cur):
        if not torch.equal(a, b):
            bad.setdefault(name, 0)
            bad[name] += 1
            break          # the first differing stage of this pass
print(f"{iters} passes at batch {n}: first differing stage per pass -> {bad if bad else 'none (all identical)'}", flush=True)

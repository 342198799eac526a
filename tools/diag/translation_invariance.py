"""Where does the HIP detector stop being translation invariant?  A constant frame gives the oracle EXACT score ties between
interior points of a level (identical receptive fields); this script runs the FCOS engine on such a frame and reports, stage
by stage, how many distinct values the interior of column 0 / of the interior block holds per channel."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
sys.path.insert(0, str(REPO))
from hn_amd import ops, synth  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402


def distinct(t, name, rows=slice(20, 70), col=0):
    """t [1,h,w,c] fp32: number of distinct rows (over all channels) among t[0, rows, col]"""
    x = t[0, rows, col].reshape(-1, t.shape[-1])
    u = torch.unique(x, dim=0).shape[0]
    spread = float((x - x[0]).abs().max())
    print(f"{name:28s} shape {tuple(t.shape)}  distinct rows in column {col}, rows {rows.start}..{rows.stop}: {u}   max spread {spread:.3e}")


def main():
    val = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
    sd = synth.make_fcos_state_dict(0, 3)
    eng = FCOSEngine(sd, 3, device="cuda")
    img = torch.full((1, 3, 480, 640), val, device="cuda")
    oh, ow, ph, pw = eng.geometry(480, 640)
    x32 = ops.fcos_preprocess(img, oh, ow, ph, pw, eng.image_mean, eng.image_std)
    print("canvas distinct values per channel (image area):", [int(torch.unique(x32[0, :oh, :ow, c]).numel()) for c in range(3)])
    x16 = ops.fcos_preprocess_split(img, oh, ow, ph, pw, eng.image_mean, eng.image_std)
    with ops.f16_terms(3):
        feats = eng.backbone(x16)
        for i, f in enumerate(feats):
            distinct(ops.from_split(f), f"FPN level {i}", rows=slice(f.shape[1] // 4, 3 * f.shape[1] // 4))
            distinct(ops.from_split(f), f"FPN level {i} (col 5)", rows=slice(f.shape[1] // 4, 3 * f.shape[1] // 4), col=5)
        outs = eng.heads(feats)
    for i, (cls_lr, reg_ctr, _) in enumerate(outs):
        h = cls_lr.shape[1]
        distinct(cls_lr, f"cls_lr level {i}", rows=slice(h // 4, 3 * h // 4))
        distinct(reg_ctr, f"reg_ctr level {i}", rows=slice(h // 4, 3 * h // 4))
        distinct(cls_lr, f"cls_lr level {i} (col 5)", rows=slice(h // 4, 3 * h // 4), col=5)
    # body stages one by one
    with ops.f16_terms(3):
        x = ops.conv_stem_pool_split(x16, eng.stem16.w16, eng.stem16.bias, 64, r=7, stride=2)
        distinct(ops.from_split(x), "stem+pool", rows=slice(50, 150))
        for bi, blk in enumerate(eng.blocks):
            o = eng._conv(x, blk["c1"], relu=True)
            idn = eng._conv(x, blk["ds"]) if blk["ds"] is not None else x
            x = eng._conv(o, blk["c2"], relu=True, residual=idn)
            h = x.shape[1]
            distinct(ops.from_split(x), f"block {bi} (layer {blk['layer']})", rows=slice(h // 4, 3 * h // 4))


if __name__ == "__main__":
    main()

"""Stem conv (7x7/2, 3->64 on the split image) timing by tile (HN_STEM_TILE experiment switch)."""
import sys, os
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops, synth
from hn_amd.fcos_engine import FCOSEngine, IMAGE_MEAN, IMAGE_STD
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)
eng = FCOSEngine(synth.make_fcos_state_dict(0, 3), 3)
rgb = synth.make_rgb(32, seed=1000).cuda()
x = ops.fcos_preprocess_split(rgb, 800, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
for _ in range(3):
    y = ops.conv_stem_split(x, eng.stem16.w16, eng.stem16.bias, 64)
torch.cuda.synchronize()
t = ops.HipTimer(); t.start()
for _ in range(50):
    y = ops.conv_stem_split(x, eng.stem16.w16, eng.stem16.bias, 64)
t.stop()
print(f"HN_STEM_TILE={os.environ.get('HN_STEM_TILE','0')}: stem {t.elapsed_ms()/50*1e3:.1f} us")
p = ops.maxpool3x3s2_nhwc(y)
t.start()
for _ in range(50):
    p = ops.maxpool3x3s2_nhwc(y)
t.stop()
print(f"   maxpool {t.elapsed_ms()/50*1e3:.1f} us")

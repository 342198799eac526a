import sys
sys.path.insert(0, "handnet-pipeline_amd")
import torch
from hn_amd import ops, synth
from hn_amd.fcos_engine import IMAGE_MEAN, IMAGE_STD
rgb = synth.make_rgb(16, seed=1000).cuda()
a = ops.fcos_preprocess_split(rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
torch.cuda.synchronize()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    b = ops.fcos_preprocess_split(rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
    torch.cuda.synchronize()
    d = (a != b)
    full = int(d.sum())
    inner = int(d[:, :, 3:-3, 3:-3, :3].sum())
    idx = torch.nonzero(d)[:4].tolist()
    vals = [(float(a[tuple(i)]), float(b[tuple(i)])) for i in idx]
    print("pass", it, "differing halfs: whole buffer", full, "interior", inner, idx, vals, flush=True)
f = ops.fcos_preprocess(rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
g = ops.fcos_preprocess(rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
print("f32 kernel equal:", torch.equal(f, g))
ops.set_form("preprocess_generic", True)
c = ops.fcos_preprocess_split(rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
e = ops.fcos_preprocess_split(rgb, 799, 1066, 800, 1088, IMAGE_MEAN, IMAGE_STD)
print("generic split kernel: equal to itself", torch.equal(c, e), "equal to tiled", torch.equal(c, a), int((c != a).sum()))

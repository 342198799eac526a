"""Where does each output element land?  1x1 identity convolution of x[m][c] = 100 m + c (development aid)."""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops
from hn_amd.weights import split_f16x3
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = 64
m = 128
x = (torch.arange(m).float()[:, None] * 100 + torch.arange(c).float()[None, :]).reshape(1, 8, 16, c)
w = torch.eye(c).reshape(c, 1, 1, c).contiguous()
y = ops.conv2d_nhwc(x.cuda(), w.cuda(), None, tile=tile, w16=split_f16x3(w).cuda()).cpu().reshape(m, c)
xx = x.reshape(m, c)
bad = (y != xx)
print("mismatches", int(bad.sum()), "of", m * c)
if bad.any():
    for r in range(0, 20):
        print(r, [f"{int(v // 100)}:{int(v % 100)}" for v in y[r, :40].tolist()])

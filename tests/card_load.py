"""A second process that keeps the card busy with f16x3 convolutions for argv[1] seconds (tests/test_dist_gpu.py:
test_stages_are_repeatable_next_to_a_second_process).  Prints `loading` once its first launches have completed."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "handnet-pipeline_amd"))
from hn_amd import ops  # noqa: E402
from hn_amd.weights import ConvW  # noqa: E402

g = torch.Generator().manual_seed(1)
x = ops.to_split(torch.randn((8, 100, 136, 256), generator=g).cuda())
cw = ConvW(torch.randn((256, 3, 3, 256), generator=g) * 0.02, None, 1, 1, 1).to("cuda")
t0, said = time.time(), False
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(50):
        ops.conv2d_nhwc(x, cw.w, None, pad=1, w16=cw.w16, out_split=True)
    torch.cuda.synchronize()
    if not said:
        print("loading", flush=True)
        said = True

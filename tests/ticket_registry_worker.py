"""Run by tests/test_conv_gpu.py in its own process (it uses up the process-wide ticket registry): split-K convolutions with more
distinct workspace addresses than the library has ticket slots.  The first kTicketSlots addresses reduce in their last
workgroups, the later ones fall back to the separate reduction launch; every result must equal the reference bit for bit, and
an early address must still work (its slot is never handed to another address)."""
import sys
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
from hn_amd import ops  # noqa: E402
from hn_amd.weights import split_f16x3  # noqa: E402

g = torch.Generator().manual_seed(5)
x = ops.to_split(torch.randn((1, 25, 34, 512), generator=g).cuda())
wt = torch.randn((512, 3, 3, 512), generator=g) * (2.0 / (512 * 9)) ** 0.5
b = torch.randn((512,), generator=g).cuda()
kw = dict(stride=1, pad=1, dil=1, relu=True, tile=7, w16=split_f16x3(wt).cuda(), force_splits=4, out_split=True)
ops.set_form("conv_no_fused_reduce", True)
ref = ops.conv2d_nhwc(x, wt.cuda(), b, **kw).clone()
ops.set_form("conv_no_fused_reduce", False)

WS_WORDS = ops.CONV_WORKSPACE_BYTES // 4
STEP = 1 << 20   # words between the workspace addresses
COUNT = 200      # > kTicketSlots (128)
big = torch.empty((WS_WORDS + COUNT * STEP,), device="cuda", dtype=torch.float32)
which = [0]
ops._conv_workspace = lambda device: big[which[0] * STEP: which[0] * STEP + WS_WORDS]
bad = 0
for i in list(range(COUNT)) + [0, 1, 127, 128, 199]:
    which[0] = i
    for _ in range(2):
        bad += 0 if torch.equal(ops.conv2d_nhwc(x, wt.cuda(), b, **kw), ref) else 1
print("mismatching launches:", bad)
sys.exit(1 if bad else 0)

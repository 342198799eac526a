"""overlap.py with CU masks: the HBM-bound GroupNorm-apply pass on a stream that owns a few reserved CUs
(hipExtStreamCreateWithCUMask), the power-capped tower conv on a stream that owns the rest.  Premise: at the power cap the
conv does not get slower on fewer CUs (the clock rises), and the pass no longer waits for conv workgroups to leave.
    python tools/probes/exp/overlap_cumask.py [reserved CUs per XCD, default 2]"""
import ctypes as C
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops
from hn_amd.weights import split_f16x3
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

per_xcd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


torch.cuda.init()
torch.zeros(1, device="cuda")
# CU i of the mask: the driver enumerates CUs XCD-interleaved (bit i -> XCD i % 8); reserve the first `per_xcd` of every XCD
small = 0
for k in range(per_xcd):
    for x in range(8):
        small |= 1 << (k * 8 + x)
full = (1 << 256) - 1
big = full & ~small

g = torch.Generator().manual_seed(0)
n, h, w = 32, 100, 136
x = ops.to_split(torch.randn((n, h, w, 256), generator=g).cuda())
wt = (torch.randn((512, 3, 3, 256), generator=g) * 0.03)
w16 = split_f16x3(wt).cuda(); wt = wt.cuda()
raw = torch.randn((n, h, w, 512), generator=g).cuda()
sc = (torch.rand((n, 512), generator=g) + 0.5).cuda(); sh = torch.randn((n, 512), generator=g).cuda() * 0.1
y = torch.empty((n, h, w, 512), device="cuda")
act = torch.empty((n, h, w, 16, 2, 32), device="cuda", dtype=torch.float16)


def conv(): ops.conv2d_nhwc(x, wt, None, pad=1, w16=w16, out=y)
def apply(): ops.to_split(raw, sc, sh, relu=True, out=act)


def run(sa, sb, fa, fb, iters=150):
    for _ in range(20):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    for _ in range(iters):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
s_big, s_small = masked_stream(big), masked_stream(small)
print(f"reserved CUs: {bin(small).count('1')} ({per_xcd} per XCD)", flush=True)
print(f"plain streams : conv {run(plain_a, plain_b, conv, None):.0f} us  apply {run(plain_a, plain_b, None, apply):.0f} us  both {run(plain_a, plain_b, conv, apply):.0f} us", flush=True)
print(f"masked streams: conv on {bin(big).count('1')} CUs {run(s_big, s_small, conv, None):.0f} us  apply on {bin(small).count('1')} CUs "
      f"{run(s_big, s_small, None, apply):.0f} us  both {run(s_big, s_small, conv, apply):.0f} us", flush=True)

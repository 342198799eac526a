"""Workgroup residency per CU from in-kernel stamps (diagnostic lib_stamps.so via HN_LIB_PATH): start / end tick and
the hardware id of every workgroup of one conv launch -> how many workgroups a CU holds on average, and how long a slot
stays empty between two workgroups.  usage: residency.py tile n h w cin cout r"""
import ctypes as C, sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import numpy as np, torch
from hn_amd import ops, _lib
from hn_amd.weights import split_f16x3
tile, n, h, w, cin, cout, r = map(int, sys.argv[1:8])
g = torch.Generator().manual_seed(0)
x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
wt = torch.randn((cout, r, r, cin), generator=g) * 0.03
w16 = split_f16x3(wt).cuda(); wt = wt.cuda(); b = torch.randn((cout,), generator=g).cuda()
for _ in range(5):
    y = ops.conv2d_nhwc(x, wt, b, pad=r // 2, relu=True, w16=w16, out_split=True, tile=tile)
torch.cuda.synchronize()
lib = _lib.load()
nb = 8192
buf = (C.c_ulonglong * (8 * nb))()
lib.hn_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.hn_debug_read_stamps(buf, 8 * nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8)
a = a[a[:, 0] > 0]
start, end = a[:, 0].astype(np.int64), a[:, 6].astype(np.int64)
hw = (a[:, 7] & 0xFFFFFFFF).astype(np.int64); xcc = (a[:, 7] >> 32).astype(np.int64)
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
print(f"{len(a)} workgroups, {len(np.unique(cu))} distinct CUs, median workgroup life {np.median(end - start):.0f} ticks")
res = []
gaps = []
for c in np.unique(cu):
    m = cu == c
    s, e = np.sort(start[m]), np.sort(end[m])
    busy = (end[m] - start[m]).sum()
    res.append(busy / (end[m].max() - start[m].min()))   # s_memtime is per XCD: spans are taken per CU
    # slot hand-over: k-th start after the first two vs (k-2)-th end (two slots)
    if len(s) > 4:
        gaps.extend((s[2:] - e[:-2]).tolist())
res = np.array(res); gaps = np.array(gaps)
print(f"average resident workgroups per CU: {res.mean():.2f} (min {res.min():.2f}, max {res.max():.2f})")
print(f"slot hand-over (next start - earlier end, two slots per CU): median {np.median(gaps):.0f} ticks, p10 {np.percentile(gaps,10):.0f}, p90 {np.percentile(gaps,90):.0f}")

# (s_memtime counters are not synchronised between CUs: only per-CU differences are meaningful)

"""The layer-walking chain kernel (csrc/conv_chain.hip) against the same convolutions as separate launches: ResNet-34 layer3 at
batch 1 -- L convolutions 256 -> 256, 3x3, on 50 x 68, every second one with the block input as S32 residual -- (1) bit-identity of
every layer's output, repeated; (2) time per chain under hipGraph replay, both ways.   usage (GPU box): python chain_probe.py [L]"""
import ctypes as C
import sys
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
from hn_amd import _lib, ops  # noqa: E402
from hn_amd.weights import split_f16x3  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 11
lib = _lib.load()
g = torch.Generator().manual_seed(3)
n, h, w, c = 1, 50, 68, 256
x0 = ops.to_split(torch.randn((n, h, w, c), generator=g).cuda())
ws, w16s, bs = [], [], []
for l in range(L):
    wt = torch.randn((c, 3, 3, c), generator=g) * (2.0 / (c * 9)) ** 0.5 * 0.7
    ws.append(wt.cuda()); w16s.append(split_f16x3(wt).cuda()); bs.append((torch.randn((c,), generator=g) * 0.1).cuda())


def separate(outs=None):
    """conv1 (ReLU) / conv2 (+ block input, ReLU) pairs, like the BasicBlocks of the layer"""
    x, block_in, ys = x0, x0, []
    for l in range(L):
        res = block_in if l % 2 == 1 else None
        y = ops.conv2d_nhwc(x, ws[l], bs[l], stride=1, pad=1, dil=1, relu=True, residual=res, out_split=True, w16=w16s[l],
                            splitk=False, out=None if outs is None else outs[l])
        ys.append(y)
        if l % 2 == 1:
            block_in = y
        x = y
    return ys


ref = separate()
# the chain on its own output buffers
ys = [torch.empty_like(r) for r in ref]
descs = (_lib.ConvDesc * L)()
vp = lambda: (C.c_void_p * L)()
ax, aw, ab, ar, ay = vp(), vp(), vp(), vp(), vp()
x, block_in = x0, x0
for l in range(L):
    d = ops.make_conv_desc(n, h, w, c, c, 3, 3, 1, 1, 1, c, 1 if l % 2 == 1 else 0)
    d.out_split, d.res_split, d.splitk, d.terms = 1, 1 if l % 2 == 1 else 0, -1, 0
    descs[l] = d
    ax[l], aw[l], ab[l], ay[l] = x.data_ptr(), w16s[l].data_ptr(), bs[l].data_ptr(), ys[l].data_ptr()
    ar[l] = block_in.data_ptr() if l % 2 == 1 else None
    if l % 2 == 1:
        block_in = ys[l]
    x = ys[l]
tb = lib.hn_conv_chain_table_bytes(L)
host = (C.c_char * tb)()
grid = C.c_int(0)
ops.check(lib.hn_conv2d_f16x3_chain_prepare(descs, ax, aw, ab, ar, ay, L, host, C.byref(grid)), "chain_prepare")
table = torch.frombuffer(host, dtype=torch.uint8).clone().cuda()
sync = torch.zeros((lib.hn_conv_chain_sync_bytes() // 4,), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
print(f"chain of {L} layers, grid {grid.value}")
bad = 0
for rep in range(20):
    for y in ys:
        y.fill_(0)
    ops.check(lib.hn_conv2d_f16x3_chain_run(table.data_ptr(), L, grid.value, sync.data_ptr(), st), "chain_run")
    for l in range(L):
        if not torch.equal(ys[l], ref[l]):
            bad += 1
status = C.c_int(0)
lib.hn_conv_chain_status(sync.data_ptr(), C.byref(status))
print(f"bit-identity: {bad} differing layer outputs in 20 runs x {L} layers; rendezvous status {status.value}; counters at rest "
      f"{int(sync.abs().sum())}")


def timed(fn, reps=300):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


outs = [torch.empty_like(r) for r in ref]
for rep in range(3):
    t_sep = timed(lambda: separate(outs))
    t_chain = timed(lambda: ops.check(lib.hn_conv2d_f16x3_chain_run(table.data_ptr(), L, grid.value, sync.data_ptr(),
                                                                   torch.cuda.current_stream().cuda_stream), "chain_run"))
    print(f"{L} separate launches {t_sep:7.1f} us   one chain launch {t_chain:7.1f} us")
lib.hn_conv_chain_status(sync.data_ptr(), C.byref(status))
print("rendezvous status at the end:", status.value)
sys.exit(1 if bad or status.value else 0)

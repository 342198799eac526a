"""Winograd F(2x2,3x3) split-domain conv prototypes: correctness against an fp64 convolution and timing against the
direct f16x3 kernel.  WINO_LIB = one of the libraries built by build.sh (default libwino_v1.so); v2 takes fp32 input.
    python tools/probes/wino/wino.py [bench | tower | iters]"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["HN_LIB_PATH"] = os.environ.get("WINO_LIB", os.path.join(HERE, "libwino_v1.so"))
V2 = "_v2" in os.environ["HN_LIB_PATH"]

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "handnet-pipeline_amd"))
from hn_amd import _lib, ops  # noqa: E402
from hn_amd.weights import split_f16x3  # noqa: E402

lib = _lib.load()
VP = C.c_void_p
raw = C.CDLL(_lib.lib_path())
raw.hn_wino_f16x3_bank_halfs.restype = C.c_int64
raw.hn_wino_f16x3_bank_halfs.argtypes = [C.c_int, C.c_int]
raw.hn_wino_f16x3_pack.argtypes = [VP, C.c_int, C.c_int, VP]
raw.hn_conv3x3_wino_f16x3.argtypes = [VP, VP, VP, VP, VP, VP, VP]


def pack(w):
    cout, _, _, cin = w.shape
    halfs = raw.hn_wino_f16x3_bank_halfs(cout, cin)
    host = np.empty(halfs, np.float16)
    wc = np.ascontiguousarray(w.cpu().numpy())
    assert raw.hn_wino_f16x3_pack(wc.ctypes.data, cout, cin, host.ctypes.data) == 0
    return torch.from_numpy(host).cuda()


def wino(xs, u, w_shape, bias=None, relu=False, residual=None, out_split=False):
    n, h, wd = xs.shape[:3]
    cout, _, _, cin = w_shape
    d = ops.make_conv_desc(n, h, wd, cin, cout, 3, 3, 1, 1, 1, cout if relu else 0, 1 if residual is not None else 0)
    d.out_split = 1 if out_split else 0
    d.res_split = 1 if residual is not None and ops.is_split(residual) else 0
    out = (torch.empty((n, h, wd, cout // 32, 2, 32), device="cuda", dtype=torch.float16) if out_split
           else torch.empty((n, h, wd, cout), device="cuda", dtype=torch.float32))
    rc = raw.hn_conv3x3_wino_f16x3(C.byref(d), _lib.ptr(xs), _lib.ptr(u), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out),
                                   ops._stream())
    assert rc == 0, lib.hn_last_error()
    return out


def check(n, h, wd, cin, cout, relu=False, res=None, out_split=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, wd, cin, generator=g)
    w = torch.randn(cout, 3, 3, cin, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    r = torch.randn(n, h, wd, cout, generator=g) if res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), padding=1)
    ref = ref.permute(0, 2, 3, 1)
    if r is not None:
        ref = ref + r.double()
    if relu:
        ref = ref.relu()
    xs = ops.to_split(x.cuda())
    xg = x.cuda() if V2 else xs
    rr = None
    if r is not None:
        rr = ops.to_split(r.cuda()) if res == "split" else r.cuda()
    y = wino(xg, pack(w), w.shape, b.cuda(), relu, rr, out_split)
    if out_split:
        y = ops.from_split(y)
    err = (y.double().cpu() - ref).abs().max().item()
    # the direct kernel on the same inputs, for scale
    w16 = split_f16x3(w).cuda()
    yd = ops.conv2d_nhwc(xs, w.cuda(), b.cuda(), pad=1, relu=relu, residual=rr, w16=w16)
    errd = (yd.double().cpu() - ref).abs().max().item()
    print(f"check n={n} {h}x{wd} {cin}->{cout} relu={relu} res={res} split={out_split}: max|err| wino {err:.3e} direct {errd:.3e}"
          f" (|ref| max {ref.abs().max().item():.2f})", flush=True)
    return err


def bench(n, h, wd, cin, cout, iters=40):
    x = torch.randn(n, h, wd, cin, device="cuda")
    w = torch.randn(cout, 3, 3, cin) / (3 * cin ** 0.5)
    b = torch.randn(cout, device="cuda")
    xs = ops.to_split(x)
    u = pack(w)
    w16 = split_f16x3(w).cuda()
    wg = w.cuda()
    flop = 2.0 * n * h * wd * cin * cout * 9

    def run(fn):
        for _ in range(40):  # the clocks settle over the first few dozen launches
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    xin = x if V2 else xs
    tw = run(lambda: wino(xin, u, w.shape, b, True))
    td = run(lambda: ops.conv2d_nhwc(xs, wg, b, pad=1, relu=True, w16=w16))
    print(f"bench n={n} {h}x{wd} {cin}->{cout}: wino {tw*1e3:.0f} us = {flop/tw/1e9:.0f} TFLOP/s | direct {td*1e3:.0f} us = "
          f"{flop/td/1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tower":
        bench(32, 100, 136, 256, 256)
        bench(32, 100, 136, 256, 256)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "iters":
        x = torch.randn(32, 100, 136, 256, device="cuda")
        w = torch.randn(256, 3, 3, 256) / 48
        xs = ops.to_split(x); u = pack(w); b = torch.randn(256, device="cuda")
        for rep in range(3):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
            ev[0].record()
            for i in range(24):
                wino(x if V2 else xs, u, w.shape, b, True)
                ev[i + 1].record()
            torch.cuda.synchronize()
            print("iters us:", " ".join(f"{ev[i].elapsed_time(ev[i+1])*1e3:.0f}" for i in range(24)), flush=True)
        sys.exit(0)
    check(1, 8, 16, 32, 128)
    check(2, 20, 30, 64, 128, relu=True)
    check(2, 25, 34, 96, 256, relu=True, res="f32")
    check(1, 13, 17, 64, 128, res="split", out_split=True)
    check(3, 100, 136, 256, 128, relu=True)
    check(1, 9, 200, 32, 128)
    if len(sys.argv) > 1 and sys.argv[1] == "bench":
        for nn in (2, 4, 8, 16, 32):
            bench(nn, 100, 136, 256, 256)
        bench(32, 100, 136, 256, 128)
        bench(32, 50, 68, 256, 256)
        bench(32, 200, 272, 64, 128)

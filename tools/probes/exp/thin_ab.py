"""A/B of the two thin-N kernels of csrc/conv3x3_thin.hip (tap tiles vs P form) and the grouped implicit GEMM on the FCOS
head-output shape (256 -> 5, five levels of an 800 x 1088 frame) over batch sizes.  One child process per form: the switch
(HN_THIN_FORM) is read once per process.   python tools/probes/exp/thin_ab.py  ->  gpurun_out/thin_ab.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SIZES = [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)]


def child(form):
    sys.path.insert(0, os.path.join(ROOT, "handnet-pipeline_amd"))
    import torch
    from hn_amd import forms, ops
    from hn_amd.weights import ConvW
    forms.apply_env()   # HN_THIN_FORM of this child
    g = torch.Generator().manual_seed(5)
    cw = ConvW(torch.randn(5, 3, 3, 256, generator=g) * 0.02, torch.randn(5, generator=g), 1, 1, 1).to("cuda")
    for n in (1, 2, 4, 8, 16, 32):
        xs = [ops.to_split(torch.randn(n, h, w, 512, generator=g).cuda())[:, :, :, :8] for h, w in SIZES]
        run = ((lambda: ops.conv2d_nhwc_grouped(xs, [cw] * 5, pad=1, relu_cols=4)) if form == "gemm"
               else (lambda: ops.conv3x3_thin_levels(xs, cw, relu_cols=4)))
        for _ in range(5):
            run()
        reps = 50
        t = ops.HipTimer()
        t.start()
        for _ in range(reps):
            run()
        t.stop()
        torch.cuda.synchronize()
        print(f"{form:5s} n={n:3d} {t.elapsed_ms() / reps * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        out = open(os.path.join(ROOT, "gpurun_out", "thin_ab.txt"), "w")
        for form in ("flat", "tap", "gemm"):
            env = dict(os.environ, HN_THIN_FORM=form)
            r = subprocess.run([sys.executable, __file__, form], env=env, capture_output=True, text=True, timeout=300)
            out.write(r.stdout + (r.stderr[-2000:] if r.returncode else ""))
            out.flush()

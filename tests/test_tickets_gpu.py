"""The in-kernel split-K reduction under the driver's eyes: `sc1` partial planes + agent-scope tickets
with concurrent launches on two host threads / streams, beside a second process on the card, and "zero at rest" checked through
the ABI (hn_debug_tickets_nonzero) after each."""
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_two_engines_on_two_threads_and_streams_equal_serial(a2j_sd):
    """Two A2JEngines (own weights, own stream => own split-K workspace => own ticket slot) run batch-1 forwards CONCURRENTLY
    from two host threads, 300 iterations each: every output equals the engine's serial output bit for bit, and no ticket
    counter is left non-zero.  Batch 1 is where every split layer reduces in its last workgroups (~33 launches per forward)."""
    from hn_amd import ops, synth
    from hn_amd.a2j_engine import A2JEngine
    assert ops.tickets_nonzero() == 0
    engines = [A2JEngine(a2j_sd, device="cuda") for _ in range(2)]
    crops = [synth.make_crops(1, 176, seed=3000 + i).cuda() for i in range(2)]
    serial = [e.forward(c).clone() for e, c in zip(engines, crops)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    wrong, errors = [0, 0], []
    start = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[i]), torch.inference_mode():
                start.wait()
                for it in range(300):
                    out = engines[i].forward(crops[i])
                    if it % 10 == 9:                      # (compare on the device: no sync per iteration)
                        wrong[i] += int(not torch.equal(out, serial[i]))
                streams[i].synchronize()
                wrong[i] += int(not torch.equal(out, serial[i]))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert wrong == [0, 0]
    assert ops.tickets_nonzero() == 0
    # the two streams really had their own workspaces (=> their own ticket slots)
    keys = {k for k in ops._WORKSPACES if k[1] in (streams[0].cuda_stream, streams[1].cuda_stream)}
    assert len(keys) == 2


def test_fused_reduction_stress_beside_a_second_process():
    """~8 s of split-K launches of every tile form that reduces in its last workgroups (2- to 16-way splits, fp32 and S32 outputs
    with residual), WHILE a second process keeps the card busy (time slicing, another XCD schedule): every result equals the
    separate-reduction reference bit for bit; afterwards every ticket is zero."""
    from hn_amd import ops
    from hn_amd.weights import split_f16x3
    load = subprocess.Popen([sys.executable, str(REPO / "tests" / "card_load.py"), "60"], stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        line = ""
        while "loading" not in line:
            line = load.stdout.readline()
            assert line or load.poll() is None, "the load process ended before it produced work"
        g = torch.Generator().manual_seed(7)
        cases = []
        for (n, h, w, cin, cout, r, stride, pad, dil), tile, splits in [
                ((1, 11, 11, 1024, 256, 1, 1, 0, 1), 7, 4), ((1, 11, 11, 512, 512, 3, 1, 2, 2), 7, 8), ((1, 25, 34, 512, 512, 3, 1, 1, 1), 3, 16),
                ((1, 50, 68, 256, 256, 3, 1, 1, 1), 12, 3), ((2, 22, 22, 512, 128, 1, 1, 0, 1), 6, 2)]:
            x = ops.to_split(torch.randn((n, h, w, cin), generator=g).cuda())
            wt = torch.randn((cout, r, r, cin), generator=g) * (2.0 / (cin * r * r)) ** 0.5
            b = torch.randn((cout,), generator=g).cuda()
            oh, ow = ops.conv_out_size(h, w, r, r, stride, pad, dil)
            res = ops.to_split(torch.randn((n, oh, ow, cout), generator=g).cuda())
            kw = dict(stride=stride, pad=pad, dil=dil, relu=True, tile=tile, w16=split_f16x3(wt).cuda(), force_splits=splits)
            run = lambda x=x, wt=wt.cuda(), b=b, res=res, kw=kw: (ops.conv2d_nhwc(x, wt, b, **kw),  # noqa: E731
                                                                ops.conv2d_nhwc(x, wt, b, residual=res, out_split=True, **kw))
            ops.set_form("conv_no_fused_reduce", True)
            try:
                ref = tuple(t.clone() for t in run())
            finally:
                ops.set_form("conv_no_fused_reduce", False)
            cases.append((run, ref))
        t0, launches, wrong = time.time(), 0, 0
        while time.time() - t0 < 8.0:
            for run, ref in cases:
                for _ in range(20):
                    out = run()
                    launches += 2
                wrong += int(not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])))
        torch.cuda.synchronize()
        assert load.poll() is None, "the load process ended before the checks did: nothing shared the card"
        assert wrong == 0 and launches > 2000, (wrong, launches)
        assert ops.tickets_nonzero() == 0
    finally:
        load.kill()
        load.wait()

"""A2J batch 64 as S crop-chunks on S streams (do the under-filled 11x11 launches of different chunks overlap?)."""
import sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
import torch
from hn_amd import ops, synth
from hn_amd.a2j_engine import A2JEngine
from hn_amd import forms as _forms
_forms.apply_env()   # development host: the HN_* A/B variables (the product never reads them)

batch = 64
a2j = A2JEngine(synth.make_a2j_state_dict(0))
x = synth.make_crops(batch, 176, seed=3000).cuda()


def run(S):
    streams = [torch.cuda.Stream() for _ in range(S)]
    chunks = x.chunk(S)

    def step():
        if S == 1:
            return [a2j.forward(x)]
        cur = torch.cuda.current_stream()
        outs = []
        for st, c in zip(streams, chunks):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(a2j.forward(c))
        for st in streams:
            cur.wait_stream(st)
        return outs
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t = ops.HipTimer(); t.start()
    for _ in range(30):
        step()
    t.stop()
    ms = t.elapsed_ms() / 30
    print(f"A2J batch {batch} on {S} stream(s): {ms:.3f} ms/step = {batch / ms * 1e3:.0f} crops/s", flush=True)


for S in (1, 2, 4):
    run(S)

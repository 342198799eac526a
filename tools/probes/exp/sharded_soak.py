"""ShardedHandNet with the step AND both RCCL collectives in one hipGraph, replayed N times on the box's single-rank RCCL group:
every gathered output compared bit for bit with the first replay's (on the device; one host read at the end).
usage (GPU box): python tools/probes/exp/sharded_soak.py [replays] [batch]"""
import os
import sys
import time
from pathlib import Path

R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from hn_amd import dist as hdist, synth  # noqa: E402
from hn_amd.a2j_engine import A2JEngine  # noqa: E402
from hn_amd.fcos_engine import FCOSEngine  # noqa: E402
from hn_amd.pipeline import HandNetEngine  # noqa: E402

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hdist.init_from_env("nccl", force=True)
torch.cuda.set_device(0)
eng = HandNetEngine(FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, device="cuda"), A2JEngine(synth.make_a2j_state_dict(0), device="cuda"), 3)
net = hdist.ShardedHandNet(eng, gather_depth=True)
rgb, depth = synth.make_rgb(n, seed=1000).cuda(), synth.make_depth(n, seed=2000).cuda()
net.prepare(rgb, depth)
print("capture:", net.capture_note, flush=True)
out = net.forward_device(rgb, depth)
torch.cuda.synchronize()
first = [t.clone() for t in (out.keypoints, out.crop_box, out.has_hand, out.valid, out.depth_rows)]
bad = torch.zeros((), device="cuda", dtype=torch.int64)
t0 = time.time()
for i in range(replays):
    out = net.forward_device(rgb, depth)
    for a, b in zip((out.keypoints, out.crop_box, out.has_hand, out.valid, out.depth_rows), first):
        bad += (a != b).sum()
    if i % 1000 == 999:
        print(f"  replay {i + 1}", flush=True)
torch.cuda.synchronize()
print(f"{replays} replays of step + record all-gather + depth all-gather at batch {n}: {int(bad)} differing values, {time.time() - t0:.1f} s "
      f"({1e3 * (time.time() - t0) / replays:.2f} ms per replay incl. the comparison)")
dist.destroy_process_group()
sys.exit(1 if int(bad) else 0)

"""Per-step cost of the N > 1 result exchange on the one GPU this box has: hn_pack_records + all_gather_into_tensor over a
single-rank RCCL group + hn_unpack_records, 32 frames per rank."""
import os, sys, time
from pathlib import Path
R = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(R / "handnet-pipeline_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch
import torch.distributed as dist
from hn_amd import dist as hdist
hdist.init_from_env("nccl", force=True)
dev = torch.device("cuda", 0)
kp = torch.rand((32, 21, 3), device=dev); box = torch.randint(0, 600, (32, 4), device=dev); has = torch.ones((32,), device=dev, dtype=torch.int32)
for _ in range(20):
    hdist.gather_results(kp, box, has, per_rank=32)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 500
for _ in range(n):
    hdist.gather_results(kp, box, has, per_rank=32)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"gather_results (pack + 1-rank RCCL all_gather_into_tensor + unpack), 32 frames: {dt * 1e6:.1f} us per step "
      f"(host-paced loop; the batch-32 step is ~28 600 us)")
dist.destroy_process_group()

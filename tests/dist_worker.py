"""Worker of tests/test_dist_gpu.py (not a test module): one rank of an N-rank run of the REAL HandNetEngine.

Launched with a torchrun-style environment (RANK / WORLD_SIZE / MASTER_*).  Every rank builds the engines from the
seeded synthetic checkpoints, takes its contiguous shard of the seeded global batch, runs forward_device and
all-gathers the per-frame records with hn_amd.dist.gather_results; rank 0 saves what it gathered.
usage: dist_worker.py <total_frames> <backend: gloo|nccl> <out.pt> [sharded]   (gloo: every rank uses cuda:0, records
travel through host memory -- the rehearsal mode of bench.py --share-gpu)"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (str(REPO), str(REPO / "handnet-pipeline_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    total, backend, out_path = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    from hn_amd import dist as hdist
    from hn_amd import synth
    from hn_amd.a2j_engine import A2JEngine
    from hn_amd.fcos_engine import FCOSEngine
    from hn_amd.pipeline import HandNetEngine
    from hn_amd import forms
    forms.apply_env()          # (test infrastructure: lets a diagnosis run select kernel forms through the HN_* variables)
    if backend == "gloo":
        os.environ["LOCAL_RANK"] = "0"
    rank, local, world = hdist.init_from_env(backend, force=True)
    # N ranks build their engines at the same time on one host: without a cap every rank's CPU ops (seeded weights, BN
    # folding, fp16 splitting) spawn a thread per core and the four-rank rehearsal took 400 s instead of 70
    torch.set_num_threads(max(1, (os.cpu_count() or 4) // (2 * world)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    eng = HandNetEngine(FCOSEngine(synth.make_fcos_state_dict(0, 3), 3, device=dev),
                        A2JEngine(synth.make_a2j_state_dict(0), device=dev), 3)
    rgb, depth = synth.make_rgb(total, seed=1000), synth.make_depth(total, seed=2000)
    if len(sys.argv) > 4 and sys.argv[4] == "sharded":
        # the product's N > 1 callable around the engine: the GLOBAL batch in, the reference's tuple out on every rank
        # (five calls: the third captures step + collectives into one hipGraph where the backend allows it)
        net = hdist.ShardedHandNet(eng, gather_depth=True)
        images = [rgb[i].to(dev) for i in range(total)]
        d = depth.to(dev)
        outs = [net(images, depth_images=d) for _ in range(5)]
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(o, outs[0])), "replayed steps differ from the eager ones"
        kp, depth_batch, crops = outs[-1]
        assert kp.device.type == "cpu" and depth_batch.is_cuda and crops.is_cuda and crops.dtype == torch.int64
        torch.save((kp, depth_batch.cpu(), crops.cpu(), world, net.gather_captured, net.capture_note), out_path + f".rank{rank}")
        dist.barrier()
        dist.destroy_process_group()
        return
    lo, hi = hdist.shard_bounds(total, rank, world)
    per_rank = -(-total // world)
    out = eng.forward_device(rgb[lo:hi].to(dev), depth[lo:hi].to(dev))
    if backend == "gloo":
        g = hdist.gather_results(out.keypoints.cpu(), out.crop_box.cpu(), out.has_hand.cpu(), per_rank=per_rank)
    else:
        g = hdist.gather_results(out.keypoints, out.crop_box, out.has_hand, per_rank=per_rank)
    kp, box, has = hdist.compact_gathered(*g)
    if rank == 0:
        torch.save((kp.cpu(), box.cpu(), has.cpu(), world), out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

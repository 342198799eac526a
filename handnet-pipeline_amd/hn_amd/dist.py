"""Multi-GPU inference: frames are independent, so a batch is sharded contiguously over the
ranks (one process per GPU, full weight replica each) and the only exchange is ONE
all-gather of fixed-size per-frame records per step (RCCL over xGMI on the GPU box, gloo
in the CPU tests).  No collective sits on the data path of the networks themselves.

Per-frame record: keypoints [21,3] fp32 + padded crop box [4] int64 + has_hand flag --
~290 bytes; 32 frames/rank => 9 KB/rank, latency-bound.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split; the first (total % world) ranks take one extra frame."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(keypoints: torch.Tensor, crop_box: torch.Tensor, has_hand: torch.Tensor,
                   per_rank: int | None = None, group=None):
    """All-gather one step's per-frame results.

    keypoints [b,J,3] fp32, crop_box [b,4] int64, has_hand [b] int32 (this rank's frames,
    b <= per_rank).  Shards are padded to `per_rank` rows so a plain all-gather suffices;
    returns (keypoints [W*per_rank,J,3], crop_box [W*per_rank,4], has_hand [W*per_rank],
    valid [W*per_rank] bool) on every rank, in global frame order.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    b = keypoints.shape[0]
    per_rank = b if per_rank is None else per_rank
    if b > per_rank:
        raise ValueError("shard larger than per_rank")
    dev = keypoints.device
    j3 = keypoints.shape[1] * keypoints.shape[2]
    fl = torch.zeros((per_rank, j3), device=dev, dtype=torch.float32)
    fl[:b] = keypoints.reshape(b, j3)
    meta = torch.zeros((per_rank, 6), device=dev, dtype=torch.int64)
    meta[:b, :4] = crop_box
    meta[:b, 4] = has_hand.to(torch.int64)
    meta[:b, 5] = 1  # row is a real frame
    if world > 1:
        fl_all = torch.empty((world * per_rank, j3), device=dev, dtype=torch.float32)
        meta_all = torch.empty((world * per_rank, 6), device=dev, dtype=torch.int64)
        dist.all_gather_into_tensor(fl_all, fl, group=group)
        dist.all_gather_into_tensor(meta_all, meta, group=group)
    else:
        fl_all, meta_all = fl, meta
    kp = fl_all.reshape(world * per_rank, keypoints.shape[1], keypoints.shape[2])
    return kp, meta_all[:, :4], meta_all[:, 4].to(torch.int32), meta_all[:, 5].bool()


def compact_gathered(kp, crop_box, has_hand, valid):
    """Drop the padding rows of gather_results -> tensors over the real global batch."""
    return kp[valid], crop_box[valid], has_hand[valid]

"""Multi-GPU inference: frames are independent, so a batch is sharded contiguously over the
ranks (one process per GPU, full weight replica each) and the only exchange is ONE
all-gather (a single collective call) of fixed-size per-frame records per step (RCCL over xGMI on the GPU box, gloo
in the CPU tests).  No collective sits on the data path of the networks themselves.

Per-frame record: keypoints [21,3] fp32 + padded crop box [4] int64 + has_hand flag --
~290 bytes; 32 frames/rank => 9 KB/rank, latency-bound.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, force: bool = False, init_method: str | None = None,
                  timeout_s: float | None = None) -> tuple[int, int, int]:
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) -> (rank, local_rank, world).
    A process group is created for world > 1, or for a single rank when `force` is set (then gather_results() still
    goes through the collective: the one-GPU rehearsal of the RCCL path).  init_method: a rendezvous of the caller's own
    (bench.py's launch ladder gives every attempt a fresh `file://` store) instead of env://; timeout_s bounds the
    rendezvous and -- through the backend's watchdog -- every collective of the group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if init_method is not None:
            kw["init_method"] = init_method
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split; the first (total % world) ranks take one extra frame."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# one fixed-size record per frame, packed so that ONE all-gather moves everything:
#   bytes   0..31   padded crop box, 4 x int64
#   bytes  32..35   has_hand (int32)        bytes 36..39  row-is-a-real-frame (int32)
#   bytes  40..     keypoints, J*3 x fp32
_HEAD = 40
_BUFFERS = {}


def _record_bytes(j3: int) -> int:
    return (_HEAD + 4 * j3 + 7) // 8 * 8


def _buffers(per_rank, world, rec, dev):
    """Send / receive buffers are reused from step to step (no allocation on the timed path)."""
    grouped = world > 1 or dist.is_initialized()
    key = (per_rank, world, rec, str(dev), grouped)
    buf = _BUFFERS.get(key)
    if buf is None:
        send = torch.zeros((per_rank, rec), device=dev, dtype=torch.uint8)
        recv = torch.zeros((world * per_rank, rec), device=dev, dtype=torch.uint8) if grouped else send
        buf = _BUFFERS[key] = (send, recv)
    return buf


def gather_results(keypoints: torch.Tensor, crop_box: torch.Tensor, has_hand: torch.Tensor,
                   per_rank: int | None = None, group=None):
    """All-gather one step's per-frame results with ONE collective.

    keypoints [b,J,3] fp32, crop_box [b,4] int64, has_hand [b] int32 (this rank's frames,
    b <= per_rank).  Shards are padded to `per_rank` records so a plain all-gather suffices;
    returns (keypoints [W*per_rank,J,3], crop_box [W*per_rank,4], has_hand [W*per_rank],
    valid [W*per_rank] bool) on every rank, in global frame order.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    b = keypoints.shape[0]
    per_rank = b if per_rank is None else per_rank
    if b > per_rank:
        raise ValueError("shard larger than per_rank")
    dev = keypoints.device
    j, three = keypoints.shape[1], keypoints.shape[2]
    j3 = j * three
    rec = _record_bytes(j3)
    send, recv = _buffers(per_rank, world, rec, dev)
    grouped = world > 1 or dist.is_initialized()
    if keypoints.is_cuda and three == 3:
        # the GPU path: ONE pack launch (shard padding included), the collective, ONE unpack launch
        from . import ops
        with ops.on_device(dev):
            ops.pack_records(keypoints.to(torch.float32).contiguous(), crop_box.to(torch.int64).contiguous(),
                             has_hand.to(torch.int32).contiguous(), per_rank, rec, out=send)
            if grouped:  # a single-rank group still runs the collective (RCCL rehearsal on one GPU)
                dist.all_gather_into_tensor(recv, send, group=group)
            kp, box, has, valid = ops.unpack_records(recv, j)
        return kp, box, has, valid.bool()
    # host tensors (gloo rehearsal, CPU tests): the same record layout with torch ops
    if b < per_rank:
        send[b:].zero_()
    send[:b, :32] = crop_box.to(torch.int64).contiguous().view(torch.uint8).reshape(b, 32)
    flags = torch.stack([has_hand.to(torch.int32), torch.ones_like(has_hand, dtype=torch.int32)], dim=1)
    send[:b, 32:_HEAD] = flags.contiguous().view(torch.uint8).reshape(b, 8)
    send[:b, _HEAD:_HEAD + 4 * j3] = keypoints.to(torch.float32).reshape(b, j3).contiguous().view(torch.uint8)
    if grouped:
        dist.all_gather_into_tensor(recv, send, group=group)
    rows = world * per_rank
    box = recv[:, :32].contiguous().view(torch.int64).reshape(rows, 4)
    fl = recv[:, 32:_HEAD].contiguous().view(torch.int32).reshape(rows, 2)
    kp = recv[:, _HEAD:_HEAD + 4 * j3].contiguous().view(torch.float32).reshape(rows, j, three)
    return kp, box, fl[:, 0].contiguous(), fl[:, 1].bool()


def compact_gathered(kp, crop_box, has_hand, valid):
    """Drop the padding rows of gather_results -> tensors over the real global batch."""
    return kp[valid], crop_box[valid], has_hand[valid]

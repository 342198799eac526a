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


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE.json config 5 as a product callable: shard -> step -> all-gather -> the reference's tuple, on every rank
# ---------------------------------------------------------------------------------------------------------------------------
CROP = 176


class ShardedOutput:
    """One sharded step's results over the GLOBAL batch, every tensor on the rank's own device (no host sync yet):
    keypoints [N,J,3] fp32, crop_box [N,4] int64, has_hand [N] int32 (0 none, 1 hand, 2 hand with a non-finite depth crop),
    range_words [W,4] int32 (every rank's f16x3 range-contract words), depth_rows [N,C,176,176] fp32 (gather_depth; else
    the [b,C,176,176] rows of this rank's own frames), host_record: pinned uint8 [W*(per_rank+1), 296] the step copies the gathered records into (valid after
    the stream is synchronised), rows: global frame index -> row of the gathered buffers, valid [N] int32: the row-is-a-real-
    frame word of each record AS IT CAME BACK through the collective (1 everywhere when every rank delivered its shard)."""

    __slots__ = ("keypoints", "crop_box", "has_hand", "valid", "range_words", "depth_rows", "host_record", "rows", "n", "per_rank",
                 "world")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class ShardedHandNet:
    """`handnet_pipeline.HandNet` over the ranks of a process group (one process per GPU, full weight replica each; SURVEY 8e,
    north_star: "batched frames shard naturally across the 8 GPUs of one node with RCCL all-gather of detections"):

        net = HandNet(args, reload_detector=True, num_classes=3, reload_a2j=True).cuda().eval()
        hn_amd.dist.init_from_env()                       # torchrun environment -> RCCL process group
        sharded = ShardedHandNet(net, gather_depth=True)
        keypoints, depth_batch, crops = sharded(images, depth_images=depth)      # the GLOBAL batch, on every rank

    Every rank runs the step on its contiguous shard (shard_bounds), packs its per-frame results into the 296-byte records of
    the single-GPU step (+ one record with its range-contract words), ONE all_gather_into_tensor moves them, and every rank
    returns the reference's tuple over the global batch (handnet_pipeline.py:107-116): keypoints [N,21,3] on the CPU,
    depth_batch [K,1|4,176,176] and crops [K,4] int64 on the rank's device, K = frames with a hand, global frame order.
    gather_depth=True adds SURVEY 8e's optional second collective for the depth crops (123 904 B per frame); without it
    depth_batch holds the crops of THIS rank's frames only.  A rank may also pass just its own shard
    (`global_batch=<frames over all ranks>`): what bench.py does, where every rank's frames are already resident.
    No collective touches the networks' data path.  The step AND its collectives are captured into one hipGraph once the
    shapes repeat (use_graph); when the backend refuses the capture the step alone is replayed and the gather is issued
    eagerly behind it (`gather_captured`, `capture_note` say which) -- never a restart, never a fallback backend.
    `net`: the drop-in HandNet, a HandNetEngine, or (tests) any object with forward_device(images, depth) -> an object with
    keypoints / crop_box / has_hand / crops_nhwc / range_flags; host tensors take the same path with torch ops (gloo)."""

    GRAPH_AFTER = 2     # eager steps with one input shape before the step + gather is captured

    def __init__(self, net, group=None, gather_depth: bool = False, use_graph=True, rgbd: bool | None = None):
        """use_graph: True -- step + collectives in ONE hipGraph once the shapes repeat; "step" -- the engine's own captured
        step is replayed and the collectives are issued eagerly behind it (what bench.py --gpus N times by default: the form
        rehearsed on real RCCL since round 3); False -- everything eager."""
        self.net, self.group, self.gather_depth = net, group, bool(gather_depth)
        self.use_graph = "step" if use_graph == "step" else bool(use_graph)
        self.rgbd = bool(getattr(net, "RGBD", False)) if rgbd is None else bool(rgbd)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._bufs = {}
        self._graphs = {}
        self._seen = {}
        self.gather_captured = None     # True / False once a capture has been attempted
        self.capture_note = "no capture attempted yet"

    # -- the engine behind `net` -------------------------------------------------------------------------------------------
    def _engine(self):
        eng = self.net.engine() if hasattr(self.net, "engine") else self.net
        return eng

    def _stage_through_host(self, t: torch.Tensor) -> bool:
        """gloo moves host memory: a rehearsal of the N-rank plumbing on GPU tensors (ranks sharing one card) stages the
        records through the host.  Not a performance mode; RCCL ("nccl") gathers device memory in place."""
        return t.is_cuda and dist.is_initialized() and dist.get_backend(self.group) == "gloo"

    def _buffers(self, per_rank, channels, dev):
        key = (per_rank, channels, str(dev))
        b = self._bufs.get(key)
        if b is None:
            rows, w = per_rank + 1, self.world
            with torch.inference_mode(False):
                b = {"send": torch.zeros((rows, 296), dtype=torch.uint8, device=dev),
                     "recv": torch.zeros((w * rows, 296), dtype=torch.uint8, device=dev),
                     "host": torch.zeros((w * rows, 296), dtype=torch.uint8, pin_memory=dev.type == "cuda")}
                # this rank's depth crops [per_rank,C,176,176]: the send buffer of the optional second collective, and what
                # depth_batch is cut from when the crops are not gathered
                b["send_d"] = torch.zeros((per_rank, channels, CROP, CROP), dtype=torch.float32, device=dev)
                if self.gather_depth:
                    b["recv_d"] = torch.zeros((w * per_rank, channels, CROP, CROP), dtype=torch.float32, device=dev)
            self._bufs[key] = b
        return b

    def _all_gather(self, recv, send):
        if not dist.is_initialized():
            recv.copy_(send)
            return
        if self._stage_through_host(send):
            r = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(r, send.cpu(), group=self.group)
            recv.copy_(r)
            return
        dist.all_gather_into_tensor(recv, send, group=self.group)

    # -- one step on this rank's shard + the collectives (static launch sequence: capturable) ---------------------------------
    def _step(self, images, depth, per_rank, bufs, total):
        eng = self._engine()
        if self.use_graph == "step" and images.is_cuda and hasattr(eng, "graphed") and not torch.cuda.is_current_stream_capturing():
            run, s_img, s_dep, out = eng.graphed(images, depth)       # (captured at the first call with these shapes)
            s_img.copy_(images)
            s_dep.copy_(depth)
            run()
            self.capture_note = "the engine's captured step is replayed; all-gather issued eagerly behind it"
        else:
            out = eng.forward_device(images, depth)
        b = out.keypoints.shape[0]
        send, recv = bufs["send"], bufs["recv"]
        if send.is_cuda:
            from . import ops
            with ops.on_device(send.device):
                ops.pack_records(out.keypoints, out.crop_box, out.has_hand, per_rank + 1, 296, out=send)     # (rows >= b: zeros)
                if getattr(out, "range_flags", None) is not None:
                    send[per_rank, :16].view(torch.int32).copy_(out.range_flags)
        else:
            send.zero_()
            send[:b, :32] = out.crop_box.to(torch.int64).contiguous().view(torch.uint8).reshape(b, 32)
            fl = torch.stack([out.has_hand.to(torch.int32), torch.ones((b,), dtype=torch.int32)], dim=1)
            send[:b, 32:_HEAD] = fl.contiguous().view(torch.uint8).reshape(b, 8)
            j3 = out.keypoints.shape[1] * 3
            send[:b, _HEAD:_HEAD + 4 * j3] = out.keypoints.to(torch.float32).reshape(b, j3).contiguous().view(torch.uint8)
            if getattr(out, "range_flags", None) is not None:
                send[per_rank, :16] = out.range_flags.to(torch.int32).contiguous().view(torch.uint8)
        self._all_gather(recv, send)
        sd = bufs["send_d"]
        crops = out.crops_nhwc                      # [b,176,176,4]: channel 0 = depth (RGB-D: the four reordered channels)
        sd[:b].copy_(crops.permute(0, 3, 1, 2) if self.rgbd else crops[..., 0].unsqueeze(1))
        if self.gather_depth:
            if b < per_rank:
                sd[b:].zero_()
            self._all_gather(bufs["recv_d"], sd)
        bufs["host"].copy_(recv, non_blocking=True)
        return self._gathered(total, per_rank, bufs)      # (unpack + row selection: part of the step, so part of its capture)

    def forward_device(self, images, depth_images, global_batch: int | None = None) -> ShardedOutput:
        """The sync-free form: runs the step + the collectives and returns ShardedOutput (device tensors over the global batch;
        its host_record is valid after the stream is synchronised).  Replayed steps return the capture's static tensors."""
        n_in = len(images)
        if global_batch is None:
            total = n_in
            lo, hi = shard_bounds(total, self.rank, self.world)
            images = images[lo:hi] if torch.is_tensor(images) else list(images)[lo:hi]
            depth_images = depth_images[lo:hi]
        else:
            total = int(global_batch)
            lo, hi = shard_bounds(total, self.rank, self.world)
            if n_in != hi - lo:
                raise ValueError(f"rank {self.rank} of {self.world} holds frames [{lo}, {hi}) of a {total}-frame batch, "
                                 f"got {n_in} frames")
        per_rank = -(-total // self.world)
        if total < self.world:      # (decided from numbers every rank knows: ALL ranks raise, none is left waiting in the collective)
            raise ValueError(f"a {total}-frame batch leaves ranks of a {self.world}-rank group without a frame")
        batch = images if torch.is_tensor(images) else torch.stack([i.float() for i in images])
        dev = batch.device
        channels = 4 if self.rgbd else 1
        bufs = self._buffers(per_rank, channels, dev)
        key = (tuple(batch.shape), tuple(depth_images.shape), per_rank, total)
        if dev.type == "cuda" and self.use_graph is True and not self._stage_through_host(batch):
            hit = self._graphs.get(key)
            if hit is None:
                self._seen[key] = self._seen.get(key, 0) + 1
                if self._seen[key] > self.GRAPH_AFTER and self.gather_captured is not False:
                    hit = self._capture(key, batch, depth_images, per_rank, bufs, total)
            if hit is not None:
                g, s_img, s_dep, out = hit
                s_img.copy_(batch)
                s_dep.copy_(depth_images)
                g.replay()
                return out          # (static tensors of the capture: overwritten by the next replay)
        return self._step(batch, depth_images, per_rank, bufs, total)

    def prepare(self, images, depth_images, global_batch: int | None = None):
        """Capture the step + collectives for these shapes NOW (instead of after GRAPH_AFTER eager steps): a caller that
        times its steps (bench.py) calls this before its warm-up.  Returns self.gather_captured."""
        for _ in range(self.GRAPH_AFTER + 1):
            self.forward_device(images, depth_images, global_batch)
        return self.gather_captured

    def _agree(self, ok: bool) -> bool:
        """True iff EVERY rank says ok (one tiny eager all-reduce over host memory on ranks > 1): a capture that works on some
        ranks only must be dropped by all of them -- a rank that replays a captured collective beside a rank that issues it
        eagerly is fine, but the ranks must go through the same NUMBER of collectives, and only a common decision keeps the
        validation replay below from being issued by some ranks and not by others."""
        if not dist.is_initialized() or self.world == 1:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend(self.group) != "gloo":
            flag = flag.to(self._engine().device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()) == 1)

    def _capture(self, key, batch, depth, per_rank, bufs, total):
        """The step and its collectives as ONE hipGraph.  The warm-up steps run the collectives eagerly first (communicator
        and buffers exist before the capture starts).  A backend that refuses to be captured leaves gather_captured = False:
        forward_device then stays eager for the collectives (the engine's own captured step still serves the launches).
        Every rank goes through the same collectives whatever happens on it: two eager warm-up steps, an agreement on "captured"
        (`_agree`), then -- only if all ranks captured -- ONE validation replay and an agreement on "replay == eager"."""
        from . import ops

        def refused(why):
            torch.cuda.synchronize()
            self.gather_captured = False
            self.capture_note = f"capture of step + all-gather refused ({why}); the gather is issued eagerly behind the step"
            return None

        with torch.inference_mode(False), torch.no_grad():
            s_img, s_dep = torch.empty_like(batch), torch.empty_like(depth)
            s_img.copy_(batch)
            s_dep.copy_(depth)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with ops.launch_cost_hidden():
                with torch.cuda.stream(side):
                    for _ in range(2):     # (a failure HERE is a failure of the eager path: it propagates, like any step's)
                        self._step(s_img, s_dep, per_rank, bufs, total)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.current_stream().synchronize()
                eager = bufs["host"].clone()             # what the last eager step gathered for these inputs
                why, g, out = None, None, None
                try:
                    if os.environ.get("HN_TEST_REFUSE_CAPTURE_RANK") == str(self.rank):     # (tests: a one-rank refusal)
                        raise RuntimeError("refused on this rank for the test")
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        out = self._step(s_img, s_dep, per_rank, bufs, total)
                except Exception as e:  # noqa: BLE001 -- the backend (or a capture-unsafe call elsewhere in the process) refused
                    torch.cuda.synchronize()
                    why = f"{type(e).__name__}: {str(e)[:200]}"
            if not self._agree(why is None):
                return refused(why or "another rank could not capture it")
            # a captured collective must deliver what the eager one did: one replay on the same inputs, compared byte for
            # byte (a backend that captures but replays something else -- or nothing -- is treated like one that refuses)
            bufs["host"].zero_()
            g.replay()
            torch.cuda.current_stream().synchronize()
            same = torch.equal(bufs["host"], eager)
            if not self._agree(same):
                return refused("the replayed step + all-gather returned other records than the eager one" if not same
                               else "on another rank the replay returned other records than the eager step")
        self.gather_captured = True
        self.capture_note = ("step + all-gather" + (" + depth all-gather" if self.gather_depth else "")
                             + " + record copy to the host captured in ONE hipGraph")
        self._graphs[key] = (g, s_img, s_dep, out)
        return self._graphs[key]

    def _gathered(self, total, per_rank, bufs) -> ShardedOutput:
        """Views / unpacked tensors of the gathered buffers (device side; no sync)."""
        recv = bufs["recv"]
        w, rows_per = self.world, per_rank + 1
        # row of the gathered buffers for global frame f: rank r's block starts at r * (per_rank + 1)
        cached = bufs.get(("rows", total))
        if cached is None:      # (index tensors are built once per batch size: no host -> device copy on the stepping path)
            rows, rows_d = [], []
            for r in range(w):
                lo, hi = shard_bounds(total, r, w)
                rows += [r * rows_per + i for i in range(hi - lo)]
                rows_d += [r * per_rank + i for i in range(hi - lo)]
            with torch.inference_mode(False):
                cached = bufs[("rows", total)] = (rows, torch.tensor(rows, dtype=torch.int64, device=recv.device),
                                                  torch.tensor(rows_d, dtype=torch.int64, device=recv.device))
        rows, idx, idx_d = cached
        if recv.is_cuda:
            from . import ops
            with ops.on_device(recv.device):
                kp, box, has, valid = ops.unpack_records(recv, 21)
        else:
            n_rows = recv.shape[0]
            box = recv[:, :32].contiguous().view(torch.int64).reshape(n_rows, 4)
            has = recv[:, 32:36].contiguous().view(torch.int32).reshape(n_rows)
            valid = recv[:, 36:40].contiguous().view(torch.int32).reshape(n_rows)
            kp = recv[:, _HEAD:_HEAD + 252].contiguous().view(torch.float32).reshape(n_rows, 21, 3)
        words = recv.view(w, rows_per, 296)[:, per_rank, :16].contiguous().view(torch.int32).reshape(w, 4)
        if self.gather_depth:   # rows of the GLOBAL batch
            depth_rows = bufs["recv_d"].index_select(0, idx_d)
        else:                   # rows of THIS rank's frames only
            lo, hi = shard_bounds(total, self.rank, w)
            depth_rows = bufs["send_d"][:hi - lo]
        return ShardedOutput(keypoints=kp.index_select(0, idx), crop_box=box.index_select(0, idx),
                             has_hand=has.index_select(0, idx), valid=valid.index_select(0, idx), range_words=words,
                             depth_rows=depth_rows,
                             host_record=bufs["host"], rows=rows, n=total, per_rank=per_rank, world=w)

    # -- the reference's tuple ---------------------------------------------------------------------------------------------
    def forward(self, images, depth_images=None, is_3D: bool = False, is_detect: bool = False, global_batch: int | None = None):
        if is_detect or is_3D:
            return None
        if depth_images is None:
            raise ValueError("depth_images is required for the ensemble inference branch")
        out = self.forward_device(images, depth_images, global_batch)
        dev = out.keypoints.device
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        import numpy as np
        a = out.host_record.numpy()
        rows = np.asarray(out.rows)
        n = out.n
        kp = torch.from_numpy(np.ascontiguousarray(a[rows, _HEAD:_HEAD + 252]).view(np.float32).reshape(n, 21, 3))
        has = torch.from_numpy(np.ascontiguousarray(a[rows, 32:36]).view(np.int32).reshape(n))
        # the f16x3 range contract over EVERY rank's words (all ranks see the same words: all raise, or none does)
        words = np.ascontiguousarray(a.reshape(out.world, out.per_rank + 1, 296)[:, out.per_rank, :16]).view(np.int32)
        words = words.reshape(out.world, 4).max(axis=0).tolist()
        noted = any(getattr(self._engine(), k, False) for k in ("note_range", "check_range"))
        if noted:
            from .pipeline import check_range_contract
            check_range_contract(kp, words, None, has_hand=has)
        mask = has != 0
        hands = int(mask.sum())
        if hands == 0:    # handnet_pipeline.py:107-108 (the placeholder has the GLOBAL batch's shape)
            shape = (n,) + tuple(depth_images.shape[1:])
            return torch.zeros((n, 21, 3)), torch.zeros(shape, dtype=depth_images.dtype, device=depth_images.device), torch.zeros((n, 4))
        idx = mask.nonzero().flatten().to(dev)
        crops = out.crop_box.index_select(0, idx)
        if self.gather_depth:
            depth_batch = out.depth_rows.index_select(0, idx)
        else:   # this rank's frames only: the rows of its own shard that hold a hand
            lo, hi = shard_bounds(n, self.rank, self.world)
            mine = (mask[lo:hi]).nonzero().flatten().to(dev)
            depth_batch = out.depth_rows.index_select(0, mine)
        return kp, depth_batch, crops

    __call__ = forward

"""N > 1 path on CPU: two gloo ranks shard a batch, run the (CPU oracle) per-frame stage on
their shard, all-gather the fixed-size records with hn_amd.dist.gather_results, and every
rank must end up with exactly the single-process result in global frame order."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_stage(frames: torch.Tensor):
    """Deterministic per-frame function standing in for the GPU pipeline (frames are independent)."""
    n = frames.shape[0]
    kp = torch.stack([torch.arange(63, dtype=torch.float32).reshape(21, 3) * 0.01 + frames[i].sum() for i in range(n)])
    box = torch.stack([torch.tensor([int(frames[i, 0] * 100), 2, 30 + int(frames[i, 1] * 50), 40], dtype=torch.int64)
                       for i in range(n)])
    has = (frames[:, 0] > 0.2).to(torch.int32)
    return kp, box, has


def _worker(rank, world, port, total, out_dir):
    for p in (str(REPO), str(REPO / "handnet-pipeline_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from hn_amd import dist as hdist
    r, _, w = hdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(123)
    frames = torch.rand((total, 4), generator=g)
    lo, hi = hdist.shard_bounds(total, rank, world)
    per_rank = -(-total // world)
    kp, box, has = _fake_stage(frames[lo:hi])
    gk, gb, gh, valid = hdist.gather_results(kp, box, has, per_rank=per_rank)
    ck, cb, ch = hdist.compact_gathered(gk, gb, gh, valid)
    torch.save((ck, cb, ch), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (8, 32), (8, 27)])
def test_n_rank_gather_equals_single_process(tmp_path, world, total):
    """World size 2 and the REAL world size 8 (BASELINE config 5: 8 ranks; 27 frames = ragged shards of 4 and 3).
    Eight GPU-touching ranks cannot be rehearsed on the one-GPU box (its process guard allows six), so the 8-rank
    rendezvous / sharding / record plumbing is covered here on gloo."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(123)
    frames = torch.rand((total, 4), generator=g)
    kp, box, has = _fake_stage(frames)
    for rank in range(world):
        ck, cb, ch = torch.load(tmp_path / f"rank{rank}.pt")
        assert torch.equal(ck, kp) and torch.equal(cb, box) and torch.equal(ch, has)
        assert cb.dtype == torch.int64 and ch.dtype == torch.int32


def test_shard_bounds_cover_batch():
    from hn_amd.dist import shard_bounds
    for total in (1, 7, 32, 255, 256):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_gather_is_identity():
    from hn_amd.dist import compact_gathered, gather_results
    kp = torch.rand((3, 21, 3))
    box = torch.randint(0, 600, (3, 4))
    has = torch.tensor([1, 0, 1], dtype=torch.int32)
    gk, gb, gh, valid = gather_results(kp, box, has, per_rank=5)
    assert gk.shape == (5, 21, 3) and valid.tolist() == [True, True, True, False, False]
    ck, cb, ch = compact_gathered(gk, gb, gh, valid)
    assert torch.equal(ck, kp) and torch.equal(cb, box) and torch.equal(ch, has)


def test_single_rank_group_still_runs_the_collective(tmp_path):
    """With a process group of one rank gather_results() goes through all_gather_into_tensor (the one-GPU RCCL
    rehearsal of tests/test_dist_gpu.py relies on it) and returns the same records."""
    import torch.distributed as dist
    from hn_amd.dist import compact_gathered, gather_results
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1)
    try:
        kp, box, has = torch.rand((3, 21, 3)), torch.randint(0, 600, (3, 4)), torch.tensor([1, 0, 1], dtype=torch.int32)
        seen = []
        real = dist.all_gather_into_tensor

        def spy(out, inp, group=None):
            seen.append(out.data_ptr() != inp.data_ptr())
            return real(out, inp, group=group)
        dist.all_gather_into_tensor = spy
        try:
            gk, gb, gh, valid = gather_results(kp, box, has, per_rank=4)
        finally:
            dist.all_gather_into_tensor = real
        assert seen == [True]
        ck, cb, ch = compact_gathered(gk, gb, gh, valid)
        assert torch.equal(ck, kp) and torch.equal(cb, box) and torch.equal(ch, has)
    finally:
        dist.destroy_process_group()


def test_gather_is_one_collective_with_reused_buffers(monkeypatch):
    """The timed path issues ONE all_gather_into_tensor per step and allocates its buffers once."""
    import torch.distributed as dist
    from hn_amd import dist as hdist
    calls = []
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)

    def fake_gather(out, inp, group=None):
        calls.append((out.data_ptr(), inp.data_ptr()))
        out[: inp.shape[0]] = inp
        out[inp.shape[0]:] = inp
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_gather)
    kp, box, has = torch.rand((3, 21, 3)), torch.randint(0, 600, (3, 4)), torch.tensor([1, 0, 1], dtype=torch.int32)
    for _ in range(3):
        gk, gb, gh, valid = hdist.gather_results(kp, box, has, per_rank=4)
    assert len(calls) == 3 and len(set(calls)) == 1
    assert gk.shape == (8, 21, 3) and torch.equal(gk[:3], kp) and torch.equal(gk[4:7], kp)
    assert torch.equal(gb[4:7], box) and gh.tolist() == [1, 0, 1, 0, 1, 0, 1, 0]
    assert valid.tolist() == [True, True, True, False] * 2


def _bench(args, env=None, timeout=600):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HN_BENCH_WORKER")}
    base.update(env or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + args, env=base, capture_output=True, text=True, timeout=timeout)


STUB = ["--stub-engine", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1", "--batch", "4", "--init-timeout", "60"]


def test_bench_eight_ranks_through_its_own_main():
    """the REAL world size of BASELINE config 5 through bench.py's own main() -- supervisor, eight fresh
    workers, file:// rendezvous, device report, the per-step all-gather, per-rank times, ONE line from rank 0 -- with a CPU
    stand-in for the engine (gloo; the one-GPU box's process guard cannot host eight GPU ranks)."""
    r = _bench(["--gpus", "8"] + STUB)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 32 and d["config"]["rccl_ranks"] == 8
    assert d["config"]["devices"] == [f"cpu-stub:{i}" for i in range(8)]
    assert "attempt 1 of 2" in d["config"]["ipc_mode"] and "HSA_ENABLE_IPC_MODE_LEGACY=0" in d["config"]["ipc_mode"]
    lo, med, hi = d["rank_ms"]["min_median_max"]
    assert len(d["rank_ms"]["per_rank"]) == 8 and lo <= med <= hi
    assert hi - lo > 2.0                 # the stub's rank 7 sleeps 3.5 ms longer than rank 0: a straggler is VISIBLE
    assert abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]


def test_bench_launch_ladder_second_rung_after_an_injected_init_failure():
    """One rank's process group "cannot start" on the first attempt (what a host with the other IPC-handle mode looks like):
    the other ranks -- blocked in the rendezvous -- are stopped by pid, ONE fresh set starts with HSA_ENABLE_IPC_MODE_LEGACY
    unset, and the line says which mode ran and why.  A failure on BOTH rungs exits non-zero with the error text, no line."""
    r = _bench(["--gpus", "4"] + STUB, env={"HN_BENCH_INJECT_INIT_FAILURE": "2:0"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    mode = d["config"]["ipc_mode"]
    assert d["n_gpus"] == 4 and d["config"]["rccl_ranks"] == 4
    assert "HSA_ENABLE_IPC_MODE_LEGACY unset" in mode and "attempt 2 of 2" in mode and "hipIpcGetMemHandle" in mode
    assert "attempt 1 failed to initialise" in r.stderr
    # under torchrun-style environments (one supervisor per rank, as the driver launches it): same ladder, shared directory
    port = _free_port()
    procs = []
    for rank in range(2):
        env = {k: v for k, v in os.environ.items() if k != "HN_BENCH_WORKER"}
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HN_BENCH_INJECT_INIT_FAILURE="1:0")
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2"] + STUB, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [l for o, _ in outs for l in o.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert "unset" in json.loads(lines[0])["config"]["ipc_mode"]
    # both rungs fail
    both = _bench(["--gpus", "2"] + STUB, env={"HN_BENCH_INJECT_INIT_FAILURE": "0:*"})
    assert both.returncode != 0, both.stdout + both.stderr
    assert "attempt 2 failed to initialise" in both.stderr and not any(l.startswith("{") for l in both.stdout.splitlines())


def test_bench_under_real_torchrun_with_a_failed_first_attempt():
    """The driver's exact launch form -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with the stub engine on gloo: every rank torchrun starts supervises the fresh worker
    of its own rank (the job directory is derived from the elastic agent's pid), the ladder's second rung runs after an injected
    first-attempt failure at rank 3, and exactly one line comes out."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HN_BENCH_WORKER")}
    env["HN_BENCH_INJECT_INIT_FAILURE"] = "3:0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", "4"] + STUB,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["rccl_ranks"] == 4 and len(d["rank_ms"]["per_rank"]) == 4
    assert "unset" in d["config"]["ipc_mode"] and "attempt 2 of 2" in d["config"]["ipc_mode"]


def test_bench_stops_the_peers_when_a_worker_fails_after_the_group_was_up():
    """A worker that fails AFTER the first collective (a refused configuration, a crash in the run) is not an initialisation
    failure: no second rung; the supervisor stops its peers at once -- they would otherwise sit in their next collective until
    the group's timeout -- and exits non-zero without a line."""
    import time
    t0 = time.time()
    r = _bench(["--gpus", "4"] + STUB, env={"HN_BENCH_INJECT_RUN_FAILURE": "2"})
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines()), r.stdout + r.stderr
    assert "attempt 2 of 2" not in r.stderr and "injected for the supervisor test" in r.stderr
    assert time.time() - t0 < 45.0          # (the STUB runs use a 60 s group timeout: the peers were stopped, not timed out)


def test_bench_fails_loudly_when_the_process_group_cannot_start():
    """`--gpus 2` under a torchrun-style environment whose rendezvous cannot complete (rank 1 of 2 with no rank 0): both rungs
    of the ladder time out (--init-timeout), bench.py exits non-zero with the backend's error text -- never a single-rank run,
    never a re-exec, no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HN_BENCH_WORKER")}
    env.update(RANK="1", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--stub-engine", "--dist-backend", "gloo",
                        "--steps", "1", "--warmup", "0", "--init-timeout", "5"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "failed to initialise" in (r.stdout + r.stderr) and not any(l.startswith("{") for l in r.stdout.splitlines())
    assert "attempt 2 of 2" in r.stderr


# ---------------------------------------------------------------------------------------------------------------------------
# ShardedHandNet (BASELINE config 5 as a product callable): the reference's tuple over the GLOBAL batch on every rank
# ---------------------------------------------------------------------------------------------------------------------------
class _StubNet:
    """CPU stand-in for the drop-in HandNet / its engine: per-frame results are a pure function of the frame (frames are
    independent), including frames without a hand and the depth crops."""
    RGBD = False
    note_range = True

    def forward_device(self, images, depth):
        import types
        n = images.shape[0]
        key = images.reshape(n, -1)[:, :4]
        kp = torch.stack([torch.arange(63, dtype=torch.float32).reshape(21, 3) * 0.01 + key[i].sum() for i in range(n)])
        box = torch.stack([torch.tensor([int(key[i, 0] * 100), 2, 30 + int(key[i, 1] * 50), 40], dtype=torch.int64) for i in range(n)])
        has = (key[:, 0] > 0.2).to(torch.int32)
        kp = kp * (has != 0).reshape(n, 1, 1)                      # zero rows for frames without a hand, as the engine writes
        crops = torch.zeros((n, 176, 176, 4))
        crops[..., 0] = depth[:, 0, :176, :176] * 2.0              # "the crop" = a function of the frame's depth map
        return types.SimpleNamespace(keypoints=kp, crop_box=box, has_hand=has, crops_nhwc=crops,
                                     range_flags=torch.zeros(4, dtype=torch.int32))


def _single_process_tuple(images, depth):
    """What handnet_pipeline.HandNet.forward returns for the whole batch (handnet_pipeline.py:107-116)."""
    out = _StubNet().forward_device(images, depth)
    mask = out.has_hand != 0
    if int(mask.sum()) == 0:
        return torch.zeros((len(images), 21, 3)), torch.zeros_like(depth), torch.zeros((len(images), 4))
    return out.keypoints, out.crops_nhwc[..., 0].unsqueeze(1)[mask], out.crop_box[mask]


def _sharded_inputs(total, no_hand=False):
    g = torch.Generator().manual_seed(321)
    images = torch.rand((total, 3, 8, 8), generator=g)
    images[:, 0, 0, 0] = 0.25 + 0.5 * images[:, 0, 0, 0]      # a hand ...
    images[1::3, 0, 0, 0] = 0.1                                  # ... except in frames 1, 4, 7, ...
    if no_hand:
        images[:, 0, 0, 0] = 0.1
    depth = torch.rand((total, 1, 180, 200), generator=g)
    return images, depth


def _sharded_worker(rank, world, port, total, out_dir, gather_depth, local_shards, no_hand):
    for p in (str(REPO), str(REPO / "handnet-pipeline_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from hn_amd import dist as hdist
    hdist.init_from_env("gloo")
    images, depth = _sharded_inputs(total, no_hand)
    net = hdist.ShardedHandNet(_StubNet(), gather_depth=gather_depth)
    outs = []
    for _ in range(2):      # (twice: the buffers are reused from step to step)
        if local_shards:
            lo, hi = hdist.shard_bounds(total, rank, world)
            outs.append(net(images[lo:hi], depth_images=depth[lo:hi], global_batch=total))
        else:
            outs.append(net(images, depth_images=depth))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # the vote that keeps a one-rank capture refusal from splitting the ranks (ShardedHandNet._capture): unanimous or nothing
    assert net._agree(True) is True and net._agree(rank != world - 1) is False and net._agree(False) is False
    torch.save(outs[-1], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,local_shards", [(2, 8, False), (2, 7, True), (8, 27, False), (8, 27, True)])
def test_sharded_handnet_returns_the_single_process_tuple_on_every_rank(tmp_path, world, total, local_shards):
    """Every rank of a 2- and an 8-rank group (27 frames: ragged shards of 4 and 3) gets the tuple the single-process call
    returns for the global batch -- keypoints (CPU, zero rows for frames without a hand), depth_batch and crops of the K
    frames with a hand in global order -- whether it passed the global batch or just its own shard."""
    mp.spawn(_sharded_worker, args=(world, _free_port(), total, str(tmp_path), True, local_shards, False), nprocs=world, join=True)
    kp, depth_batch, crops = _single_process_tuple(*_sharded_inputs(total))
    assert 0 < crops.shape[0] < total            # (the case holds frames with and without a hand)
    for rank in range(world):
        g_kp, g_depth, g_crops = torch.load(tmp_path / f"rank{rank}.pt")
        assert torch.equal(g_kp, kp) and g_kp.dtype == torch.float32
        assert torch.equal(g_crops, crops) and g_crops.dtype == torch.int64
        assert torch.equal(g_depth, depth_batch) and tuple(g_depth.shape[1:]) == (1, 176, 176)


def test_sharded_handnet_without_depth_gather_and_without_hands(tmp_path):
    """gather_depth=False: ONE collective; depth_batch holds the crops of the rank's OWN frames with a hand.  A batch without
    any hand returns the reference's placeholder triple over the global batch (handnet_pipeline.py:107-108)."""
    from hn_amd.dist import shard_bounds
    world, total = 2, 7
    mp.spawn(_sharded_worker, args=(world, _free_port(), total, str(tmp_path), False, False, False), nprocs=world, join=True)
    images, depth = _sharded_inputs(total)
    kp, depth_batch, crops = _single_process_tuple(images, depth)
    has = _StubNet().forward_device(images, depth).has_hand != 0
    for rank in range(world):
        g_kp, g_depth, g_crops = torch.load(tmp_path / f"rank{rank}.pt")
        lo, hi = shard_bounds(total, rank, world)
        before, mine = int(has[:lo].sum()), int(has[lo:hi].sum())
        assert torch.equal(g_kp, kp) and torch.equal(g_crops, crops)
        assert torch.equal(g_depth, depth_batch[before:before + mine])
    mp.spawn(_sharded_worker, args=(world, _free_port(), total, str(tmp_path), True, True, True), nprocs=world, join=True)
    for rank in range(world):
        g_kp, g_depth, g_crops = torch.load(tmp_path / f"rank{rank}.pt")
        assert tuple(g_kp.shape) == (total, 21, 3) and not g_kp.any()
        assert tuple(g_depth.shape) == (total, 1, 180, 200) and not g_depth.any()
        assert tuple(g_crops.shape) == (total, 4) and g_crops.dtype == torch.float32 and not g_crops.any()


def test_sharded_handnet_single_process_and_collective_count(monkeypatch):
    """Without a process group the callable is the single-process call; with one it issues exactly one collective per step
    (two with gather_depth) and reuses its buffers."""
    import torch.distributed as dist
    from hn_amd import dist as hdist
    images, depth = _sharded_inputs(5)
    want = _single_process_tuple(images, depth)
    got = hdist.ShardedHandNet(_StubNet(), gather_depth=True)(images, depth_images=depth)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    assert hdist.ShardedHandNet(_StubNet())(images, depth_images=depth, is_detect=True) is None
    calls = []
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 1)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: 0)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")

    def fake_gather(out, inp, group=None):
        calls.append((out.data_ptr(), inp.data_ptr()))
        out.copy_(inp)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_gather)
    for gather_depth, per_step in ((False, 1), (True, 2)):
        calls.clear()
        net = hdist.ShardedHandNet(_StubNet(), gather_depth=gather_depth)
        for _ in range(3):
            got = net(images, depth_images=depth)
        assert len(calls) == 3 * per_step and len(set(calls)) == per_step
        assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2])
    with pytest.raises(ValueError, match="holds frames"):
        hdist.ShardedHandNet(_StubNet())(images, depth_images=depth, global_batch=9)
    # fewer frames than ranks: refused on EVERY rank from numbers all of them know (no rank is left in the collective)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 8)
    for rank in (0, 7):
        monkeypatch.setattr(dist, "get_rank", lambda group=None, r=rank: r)
        with pytest.raises(ValueError, match="without a frame"):
            hdist.ShardedHandNet(_StubNet())(images, depth_images=depth)

"""Multi-process layer on CPU (gloo, world_size 2): utterance sharding needs no data-path collective;
the only collective is the start-up broadcast of the packed weight arena."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from fish_speech_amd.dist import owner_of, shard_utterances


def test_shard_utterances_partition():
    items = list(range(11))
    for world in (1, 2, 3, 8):
        parts = [shard_utterances(items, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == items
        for r, p in enumerate(parts):
            assert all(owner_of(i, world) == r for i in p)
    with pytest.raises(ValueError):
        shard_utterances(items, 3, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from fish_speech_amd.dist import broadcast_buffer, gather_results, shard_utterances

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. arena broadcast in chunks (here 1000-byte pieces of a 10 007-byte buffer)
        g = torch.Generator().manual_seed(5)
        ref = torch.randint(0, 256, (10007,), dtype=torch.uint8, generator=g)
        buf = ref.clone() if rank == 0 else torch.zeros_like(ref)
        broadcast_buffer(buf, src=0, chunk_bytes=1000)
        ok_bcast = bool(torch.equal(buf, ref))
        # 2. each rank "generates" its own utterances r::R; results gather back in global order
        n = 7
        mine = shard_utterances(list(range(n)), rank, world)
        local = [torch.full((3,), i, dtype=torch.int64) for i in mine]
        full = gather_results(local, world, rank)
        ok_gather = [int(t[0]) for t in full] == list(range(n))
        q.put((rank, ok_bcast, ok_gather))
    finally:
        dist.destroy_process_group()


def test_two_process_broadcast_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res


def test_lpt_partition_of_mixed_length_utterances():
    """Config 4 (64 mixed-length utterances over 8 GPUs): greedy longest-first packing keeps the heaviest rank
    within one utterance of the mean; every utterance lands on exactly one rank."""
    import random

    from fish_speech_amd.scheduler import lpt_order, partition_for_ranks

    rng = random.Random(4)
    costs = [rng.randint(100, 430) for _ in range(64)]
    bins = partition_for_ranks(costs, 8)
    assert sorted(i for b in bins for i in b) == list(range(64))
    loads = [sum(costs[i] for i in b) for b in bins]
    assert max(loads) - sum(loads) / 8 <= max(costs)
    assert max(loads) <= 1.06 * sum(loads) / 8
    order = lpt_order(costs)
    assert [costs[i] for i in order] == sorted(costs, reverse=True)
    assert partition_for_ranks([5.0], 4) == [[0], [], [], []]


# ------------------------------------------------------------------ bench.py's multi-rank logic without GPUs


class _FakeArenaModel:
    """What dist.broadcast_arena touches on MiDualAR / MiDAC: `.arena` (flat uint8) and `weights_ready()`."""

    def __init__(self, loaded: bool, n=300_001):
        g = torch.Generator().manual_seed(11)
        ref = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g)
        self.ref = ref
        self.arena = ref.clone() if loaded else torch.zeros_like(ref)
        self.ready_calls = 0
        self.loaded = loaded

    def weights_ready(self):
        assert torch.equal(self.arena, self.ref), "weights_ready() before the arena arrived"
        self.ready_calls += 1


class _StubDualAR:
    """Slot API of MiDualAR (prefill / decode / poll_done / read / release) with a deterministic fake generator on
    the CPU: frame f of an utterance = (seed + f) % 1000 in every row.  Lets scheduler.generate_queue and
    bench.run_config4 run under gloo."""

    def __init__(self, cfg, max_batch):
        self.config, self.max_batch_size, self._cache_setup_done = cfg, max_batch, True
        self.slots = {}

    def _sampling(self, t, p, k, seed, ras):
        return seed

    def next_seed(self):
        return 0

    def prefill(self, slots, prompts, max_new, samp):
        for s, p, m, seed in zip(slots, prompts, max_new, samp):
            assert s not in self.slots
            self.slots[s] = dict(seed=seed, limit=m, n=1)

    def decode(self, slots, n_frames):
        for s in slots:
            st = self.slots[s]
            st["n"] = min(st["limit"], st["n"] + n_frames)

    def poll_done(self, slots):
        return [2 if self.slots[s]["n"] >= self.slots[s]["limit"] else 0 for s in slots]

    def read(self, slot):
        st = self.slots[slot]
        ncb1 = self.config.num_codebooks + 1
        f = (st["seed"] + torch.arange(st["n"])) % 1000
        return f.view(-1, 1).expand(-1, ncb1).to(torch.int32).contiguous(), 2

    def release(self, slot):
        del self.slots[slot]


def _bench_worker(rank, world, port, q):
    import torch.distributed as dist

    import bench
    from fish_speech_amd.dist import broadcast_arena

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _FakeArenaModel(loaded=(rank == 0))
        broadcast_arena(m, src=0, chunk_bytes=65536)          # the real code path, 5 pieces
        ok_arena = bool(torch.equal(m.arena, m.ref)) and m.ready_calls == (0 if rank == 0 else 1)
        cfg = bench.s2_pro_config()
        out = bench.run_config4(_StubDualAR(cfg, bench.BATCH), None, cfg, rank, world, dist, torch.device("cpu"))
        prompts, frames = bench.mixed_length_workload(cfg, bench.BATCH * world)
        q.put((rank, ok_arena, out, sum(frames)))
    finally:
        dist.destroy_process_group()


def test_bench_rank_logic_and_arena_broadcast_under_gloo():
    """bench.py's config-4 leg (LPT partition over ranks, continuous batching per rank, MAX/SUM reductions) and
    dist.broadcast_arena + weights_ready, executed by two gloo processes with stand-in model objects."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "arena broadcast / weights_ready"
    a, b = res[0][2], res[1][2]
    assert a["frames_total"] == b["frames_total"] == res[0][3]      # every utterance ran on exactly one rank
    assert a["wall_s"] == b["wall_s"] and a["rank_imbalance_max_over_mean"] >= 1.0


def test_generate_queue_per_utterance_lengths_with_the_stub():
    import bench
    from fish_speech_amd.scheduler import generate_queue

    cfg = bench.s2_pro_config()
    prompts, frames = bench.mixed_length_workload(cfg, 11)
    res = generate_queue(model=_StubDualAR(cfg, 3), prompts=prompts, max_new_tokens=frames, max_batch=3,
                         seeds=list(range(11)), poll_every=16)
    for i, (r, p, f) in enumerate(zip(res, prompts, frames)):
        assert r.shape == (cfg.num_codebooks + 1, p.shape[1] + f)
        assert torch.equal(r[0, p.shape[1]:], (i + torch.arange(f)) % 1000)


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus 2` without a launcher self-spawns; on a box with fewer devices it must fail loudly
    instead of printing an n_gpus = 1 line."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout

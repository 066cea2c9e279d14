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

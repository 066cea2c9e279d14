"""Multi-GPU layer of the hot path: utterances are independent, so the path shards by
utterance with NO data-path collective -- rank r takes utterances r::R, the scheme of the reference's
tools/vqgan/extract_vq.py:161-207 (files[RANK::WORLD_SIZE]).  The one collective is a start-up
broadcast of the packed weight arena over RCCL/xGMI, so the checkpoint is read from disk once.

One process per GPU under torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, TypeVar

import torch

T = TypeVar("T")


def shard_utterances(items: Sequence[T], rank: int, world: int) -> List[T]:
    """Rank r of `world` owns items r, r+world, ... (extract_vq.py:207)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0,{world})")
    return list(items[rank::world])


def owner_of(index: int, world: int) -> int:
    return index % world


def broadcast_buffer(buf: torch.Tensor, src: int = 0, chunk_bytes: int = 1 << 30, group=None):
    """Broadcast a flat uint8 buffer in <=1 GiB pieces (xGMI rings are per-link bound: a few large
    pipelined pieces saturate them; one 9 GB call would also need a 32-bit-safe element count)."""
    import torch.distributed as dist

    flat = buf.view(-1)
    n = flat.numel()
    if flat.is_cuda and "nccl" not in str(dist.get_backend(group)):
        # A non-RCCL backend (gloo: the CPU tests and the one-GPU multi-rank debug mode) is handed HOST tensors, staged
        # here with synchronous copies.  gloo does accept device tensors -- it stages them through pinned buffers and
        # copies on streams of its own -- and that path, between ranks that share one GPU, is what the GPU memory access
        # faults of profiles/r06_startup_order_stress.txt need: ranks that skip the broadcast never fault.
        host = torch.empty(min(n, chunk_bytes), dtype=torch.uint8)
        rank = dist.get_rank(group)
        for off in range(0, n, chunk_bytes):
            piece = flat[off: min(n, off + chunk_bytes)]
            h = host[: piece.numel()]
            if rank == src:
                h.copy_(piece)                      # D2H, returns when done
            dist.broadcast(h, src=src, group=group)
            if rank != src:
                piece.copy_(h)                      # H2D from pageable memory: returns when the bytes have left the host
        return
    for off in range(0, n, chunk_bytes):
        dist.broadcast(flat[off: min(n, off + chunk_bytes)], src=src, group=group)


def broadcast_arena(model, src: int = 0, group=None, chunk_bytes: int = 1 << 30):
    """Replicate a loaded model's packed weight arena to every rank, then mark it ready there.

    Ordering is the product's, not the caller's.  (1) A blocking `dist.broadcast` returns with torch's current stream
    ordered after the collective (RCCL / gloo run it on their own streams), and `weights_ready()` makes the model's
    private stream -- on which the first prefill derives the row-balanced copies and the q|k|v table FROM the arena --
    wait for the current stream.  (2) The call ends with a device synchronize on every rank -- defensive: start-up is
    untimed, and it leaves every rank quiet before anybody's first step.  (The GPU memory faults of the one-GPU
    multi-rank debug mode that this was first added against turned out to belong to gloo's device-tensor broadcast, see
    broadcast_buffer and profiles/r06_startup_order_stress.txt.)  Callers need no synchronize of their own."""
    import torch.distributed as dist

    broadcast_buffer(model.arena, src=src, chunk_bytes=chunk_bytes, group=group)
    if dist.get_rank(group) != src:
        model.weights_ready()
    if model.arena.is_cuda:
        torch.cuda.synchronize(model.arena.device)


def gather_results(local: List[torch.Tensor], world: int, rank: int, group=None):
    """Collect per-utterance results on every rank in the original order (host-side; not timed)."""
    import torch.distributed as dist

    gathered: List[list] = [None] * world  # type: ignore
    dist.all_gather_object(gathered, [t.cpu() for t in local], group=group)
    total = sum(len(g) for g in gathered)
    out = [None] * total
    for r, g in enumerate(gathered):
        for j, t in enumerate(g):
            out[r + j * world] = t
    return out

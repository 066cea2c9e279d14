"""Continuous batching of a queue of utterances over the slots of one GPU (BASELINE.json config 4: mixed-length
prompts).  The reference serves one utterance at a time (`launch_thread_safe_queue`, batch 1,
fish_speech/models/text2semantic/inference.py:748-799); here up to `max_batch` utterances share every frame's
weight stream, and a slot whose utterance emitted <|im_end|> (or used up its frames) is refilled from the queue
at the next poll while the others keep decoding -- throughput is not held back by the longest utterance of a
static batch.

Determinism contract: utterance i's result depends only on (prompt_i, sampling parameters, seed_i) -- the
kernels are batch-invariant and the sampler's random stream is keyed by the seed, not by the slot -- so it
equals `generate(model, prompt_i, seed=seed_i)` whatever the arrival order or slot (tests/test_stream_gpu.py)."""
from __future__ import annotations

import numbers

from typing import List, Optional, Sequence

import torch

from .dual_ar import MiDualAR


def lpt_order(costs: Sequence[float]) -> List[int]:
    """Longest-processing-time-first admission order (indices): with E[frames] as the cost the tail of the
    schedule is short utterances, which keeps the slots busy to the end."""
    return sorted(range(len(costs)), key=lambda i: (-costs[i], i))


def partition_for_ranks(costs: Sequence[float], world: int) -> List[List[int]]:
    """Greedy LPT bin packing of utterances over ranks (one process per GPU, no collective): heaviest first,
    each to the currently lightest rank.  Returns the utterance indices per rank."""
    bins: List[List[int]] = [[] for _ in range(world)]
    load = [0.0] * world
    for i in lpt_order(costs):
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += costs[i]
    return bins


@torch.no_grad()
def generate_queue(*, model: MiDualAR, prompts: Sequence[torch.Tensor], max_new_tokens,
                   max_batch: Optional[int] = None, poll_every: int = 16, seeds: Optional[Sequence[int]] = None,
                   order: Optional[Sequence[int]] = None, temperature: float = 1.0, top_p: float = 0.9,
                   top_k: int = 30, use_ras: bool = True, stats: Optional[dict] = None) -> List[torch.Tensor]:
    """Run all `prompts` through `max_batch` slots with refill; returns, per utterance (in input order),
    (1+ncb, T_i + n_i) like the reference's `generate`.  `max_new_tokens`: one int, or one per utterance."""
    cfg = model.config
    n = len(prompts)
    if max_new_tokens is None:                          # None / 0 = no limit, like generate()
        max_new_tokens = 0
    per_utt = ([int(max_new_tokens)] * n if isinstance(max_new_tokens, numbers.Integral) or
               (isinstance(max_new_tokens, torch.Tensor) and max_new_tokens.ndim == 0)
               else [int(m) if m is not None else 0 for m in max_new_tokens])
    assert len(per_utt) == n
    for p in prompts:
        if p.size(1) >= cfg.max_seq_len:  # inference.py:263-266
            raise ValueError(f"Input sequence length {p.size(1)} exceeds max_seq_len {cfg.max_seq_len}")
    if not model._cache_setup_done:
        model.setup_caches(max_batch_size=max_batch or min(n, 8), max_seq_len=cfg.max_seq_len)
    B = min(max_batch or model.max_batch_size, model.max_batch_size)
    seeds = list(seeds) if seeds is not None else [model.next_seed() for _ in range(n)]
    pending = list(order) if order is not None else list(range(n))
    assert sorted(pending) == list(range(n)), "order must be a permutation of the utterances"
    pending.reverse()                                   # pop() takes the next one
    free = list(range(B - 1, -1, -1))
    active = {}                                         # slot -> utterance index
    results: List[Optional[torch.Tensor]] = [None] * n
    frames_run = 0
    while pending or active:
        new_slots, new_idx = [], []
        while pending and free:
            new_slots.append(free.pop())
            new_idx.append(pending.pop())
        if new_slots:
            ps = [prompts[i] for i in new_idx]
            mn = [min(per_utt[i] if per_utt[i] else cfg.max_seq_len - p.size(1), cfg.max_seq_len - p.size(1))
                  for i, p in zip(new_idx, ps)]
            samp = [model._sampling(temperature, top_p, top_k, seeds[i], use_ras) for i in new_idx]
            model.prefill(new_slots, ps, mn, samp)
            active.update(zip(new_slots, new_idx))
        slots = sorted(active)
        model.decode(slots, poll_every)
        frames_run += poll_every
        for s, d in zip(slots, model.poll_done(slots)):
            if d:
                i = active.pop(s)
                frames, _ = model.read(s)
                p = prompts[i]
                seq = torch.cat([p.to("cpu", torch.int64), frames.t().to(torch.int64)], dim=1)
                results[i] = seq.to(p.dtype) if p.dtype in (torch.int32, torch.int64) else seq
                model.release(s)
                free.append(s)
    if stats is not None:
        stats["frames_run"] = frames_run
    return results  # type: ignore[return-value]

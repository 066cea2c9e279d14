"""Mutation self-check of the Dual-AR parity fixtures (VERDICT r04 #4a): can the checks FAIL for the right reasons?

A fixture family "detects" a numeric fault if, with the fault injected into the CPU oracle, at least one of the checks the
GPU suite runs on that family fails:
  * tokens -- the free-running token matrix differs from the fixture's (what the full-sequence-equality tests assert);
  * taps   -- the teacher-forced float taps (slow logits, hidden, fast logits of every frame) leave the tolerances of
              tests/helpers.check_teacher_forced that the GPU tests apply to the family (16 bf16 steps, relative L2 2 %),
              or a decision leaves the reference's near-argmax set.
The faults are arithmetic ones an implementation could really have: a mis-scaled FFN down-projection (every layer / one
layer), RoPE not applied, an attention block that contributes nothing (fast / slow).  Shared by tests/test_oracle_cpu.py
(asserts detection) and tools/mutation_table.py (writes profiles/r05_mutation_table.txt)."""
from __future__ import annotations

import numpy as np
import torch

from oracle import dual_ar as O
from tests.helpers import check_teacher_forced, load_dualar_case, oracle_step_fn


def _scale(state, pred, f):
    out = dict(state)
    n = 0
    for k, v in state.items():
        if pred(k):
            out[k] = (v.float() * f).to(v.dtype)
            n += 1
    assert n > 0
    return out


def mutate_state(name, cfg, state):
    """-> (state', oracle hook or None)"""
    if name == "ffn_w2_half":          # every FFN down-projection (slow and fast) scaled by 0.5
        return _scale(state, lambda k: k.endswith("feed_forward.w2.weight"), 0.5), None
    if name == "one_layer_w2_half":    # ONE slow layer's down-projection only
        mid = f"layers.{cfg.n_layer // 2}.feed_forward.w2.weight"
        return _scale(state, lambda k: k == mid, 0.5), None
    if name == "zero_fast_attention":  # the fast transformer's attention blocks contribute nothing
        return _scale(state, lambda k: k.startswith("fast_layers.") and k.endswith("attention.wo.weight"), 0.0), None
    if name == "zero_slow_attention":
        return _scale(state, lambda k: k.startswith("layers.") and k.endswith("attention.wo.weight"), 0.0), None
    if name == "no_rope":              # rotary embedding not applied (cos = 1, sin = 0), both transformers
        def hook(orc):
            for tab in (orc.freqs, orc.fast_freqs):
                tab[..., 0] = 1.0
                tab[..., 1] = 0.0
        return state, hook
    raise KeyError(name)


MUTATIONS = ("ffn_w2_half", "one_layer_w2_half", "no_rope", "zero_fast_attention", "zero_slow_attention")


def _oracle(cfg, state, hook):
    orc = O.DualAROracle(cfg, state)
    if hook:
        hook(orc)
    return orc


def detect(case: str, mutation: str | None):
    """-> dict(tokens_changed, first_token_mismatch_frame, taps_fail, taps_reason): what the two kinds of check see when
    the oracle carries `mutation` (None = the unmodified oracle: nothing may fire)."""
    cfg, state, z = load_dualar_case(case)
    hook = None
    if mutation:
        state, hook = mutate_state(mutation, cfg, state)
    want = z["tokens"] if "tokens" in z.files else z["greedy"]
    T = z["prompt"].shape[1]
    top_k = int(z["top_k"]) if "top_k" in z.files else 1
    temperature = float(z["temperature"]) if "temperature" in z.files else 0.7
    top_p = float(z["top_p"]) if "top_p" in z.files else 0.7
    y = O.generate(_oracle(cfg, state, hook), torch.from_numpy(z["prompt"]), int(z["max_new"]), temperature, top_p, top_k,
                   uniform_fn=O.FmiUniform(int(z["uniform_seed"]), 0)).numpy()
    if y.shape == want.shape:
        bad = np.argwhere((y != want).any(axis=0)).reshape(-1)
        changed = int((y != want).sum())
        first = int(bad[0]) - T if len(bad) else None
    else:
        changed, first = -1, min(y.shape[1], want.shape[1]) - T
    res = dict(tokens_changed=changed, first_token_mismatch_frame=first, taps_fail=None, taps_reason="")
    if "slow_logits_live" in z.files:   # the teacher-forced tap check (greedy: near-argmax decisions; sampled: equal draws)
        try:
            check_teacher_forced(oracle_step_fn(cfg, state, int(z["uniform_seed"]), temperature, top_p, top_k, hook=hook), cfg, z,
                                 decide="near_argmax" if top_k == 1 else "equal")
            res["taps_fail"] = False
        except AssertionError as e:
            res["taps_fail"], res["taps_reason"] = True, str(e)[:90]
    return res

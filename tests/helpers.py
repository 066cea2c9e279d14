"""Shared helpers for the tests: golden-case loading and tolerant bf16 comparison."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import dual_ar as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_dualar_case(name: str):
    z = np.load(os.path.join(GOLDEN, f"dualar_{name}.npz"))
    kw = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        kw[str(k)] = int(v)
    cfg = O.DualARConfig(**kw)
    if "state_kind" in z.files and str(z["state_kind"]) == "peaky":   # well-conditioned fixtures (make_peaky_state)
        import json

        skw = json.loads(str(z["state_kwargs"]))
        if "hot" in skw:
            skw["hot"] = tuple(skw["hot"])
        state = O.make_peaky_state(cfg, **skw)
    else:
        state = O.make_synthetic_state(cfg, seed=int(z["state_seed"]), head_gain=float(z["head_gain"]))
    return cfg, state, z


PEAKY_GREEDY = ["tiny_peaky", "tiny_peaky_eos", "mid_peaky", "tiny_projin"]   # tiny_projin: fast_dim != dim (fast_project_in)
PEAKY_ALL = PEAKY_GREEDY + ["tiny_sampled"]


def bf16_from_u16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def bf16_close(a: torch.Tensor, b: torch.Tensor, ulps: float = 2.0, scale=None, rms_floor: bool = True):
    """Elementwise |a-b| <= ulps * ulp_bf16(m), m = max(|a|, |b|[, |scale|][, rms(b)]).

    bf16 carries 8 significant bits; the HIP path and the CPU path round the same fp32 sums taken in a
    different order, so a value may land one bf16 step away, two after a product of two such values.
    ``scale`` = magnitude of the operands when the result is a sum that may cancel (residual adds);
    ``rms_floor`` keeps the tolerance from collapsing on outputs that are themselves cancellations of
    O(rms)-sized dot-product terms.  Returns (ok, max_abs_err, n_bad)."""
    from oracle.dual_ar import bf16_ulp

    a, b = a.float().cpu(), b.float().cpu()
    mag = torch.maximum(a.abs(), b.abs())
    if scale is not None:
        mag = torch.maximum(mag, scale.float().cpu().abs())
    if rms_floor:
        fin = b[torch.isfinite(b)]
        if fin.numel():
            mag = torch.maximum(mag, fin.pow(2).mean().sqrt().expand_as(mag))
    tol = ulps * bf16_ulp(mag)
    both_inf = torch.isinf(a) & torch.isinf(b) & (a == b)
    bad = ((a - b).abs() > tol) & ~both_inf
    err = (a - b).abs()
    err[both_inf] = 0
    return (not bool(bad.any())), float(err.max()), int(bad.sum())


# --------------------------------------------------------------------------------------------------
# Teacher-forced parity of one Dual-AR implementation against the REFERENCE's stored traces.
# --------------------------------------------------------------------------------------------------


def check_teacher_forced(step_fn, cfg, z, ulps: float = 16.0, decide_ulps: float = 8.0, rel_l2: float = 2e-2,
                         decide: str = "near_argmax"):
    """step_fn(frame, x (S,1+ncb) int tensor, pos0, prev_window or None) ->
           (tokens (1+ncb,), slow_logits_live (n_live,), hidden (dim,), fast_logits (ncb-1, cbs))

    Drives the implementation with the reference's own greedy token history (tests/golden) and, for
    every frame, requires
      * every floating-point tap within ``ulps`` bf16 steps (max) and ``rel_l2`` relative L2 error of
        the reference's -- the taps sit behind 2..36 bf16 layers whose roundings differ with the fp32
        summation order, so the bound is a few steps, not one (the reference's own CPU-to-CPU spread
        is of the same size: tests/golden was written on another CPU than the GPU box's);
      * every decision (top_k=1) to land on a token whose REFERENCE logit is within ``decide_ulps``
        bf16 steps of the reference's maximum -- i.e. bit-exact indices wherever the reference's own
        margin exceeds the rounding noise of a different fp32 summation order, and never an outlier;
      * exact equality with the reference token when that token came from the u == 0 quirk of the
        exponential race (inference.py:43-46), which does not depend on the logits.
    decide="equal" (SAMPLED fixtures: the step function draws with the fixture's top-k / top-p / temperature / seed, and
    the fixture's draws are invariant under two bf16 steps of logit noise): every decision must EQUAL the reference's
    token -- the taps are checked the same way.
    Returns statistics for reporting."""
    from oracle import dual_ar as O

    seq = torch.from_numpy(z["greedy"] if "greedy" in z.files else z["tokens"])
    prompt = torch.from_numpy(z["prompt"])
    T = prompt.shape[1]
    ncb1 = cfg.num_codebooks + 1
    ids = torch.from_numpy(z["live_ids"]).long()
    ref_slow = bf16_from_u16(z["slow_logits_live"])
    ref_hidden = bf16_from_u16(z["hidden"])
    ref_fast = bf16_from_u16(z["fast_logits"])
    window = torch.zeros(ncb1, 10, dtype=torch.int32)
    stats = dict(frames=0, decisions=0, exact=0, max_ulp_slow=0.0, max_ulp_fast=0.0, max_rel_l2=0.0, miss_margins=[])

    def ulp_err(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        fin = torch.isfinite(b)
        m = torch.maximum(torch.maximum(a.abs(), b.abs()), b[fin].pow(2).mean().sqrt())
        rel = float((a - b)[fin].norm() / b[fin].norm().clamp_min(1e-20))
        stats["max_rel_l2"] = max(stats["max_rel_l2"], rel)
        assert rel <= rel_l2, f"relative L2 error {rel:.4f} > {rel_l2}"
        return float(((a - b).abs() / O.bf16_ulp(m))[fin].max())

    def near_argmax(ref_logits, picked_idx, ref_idx):
        lf = ref_logits.float()
        top = lf.max()
        if picked_idx != ref_idx:   # how far below the reference's maximum the picked token sits, in bf16 steps
            stats["miss_margins"].append(round(float((top - lf[picked_idx]) / O.bf16_ulp(top)), 2))
        return float(lf[picked_idx]) >= float(top - decide_ulps * O.bf16_ulp(top)) or picked_idx == ref_idx

    n_frames = seq.shape[1] - T
    for f in range(n_frames):
        if f == 0:
            x, pos0, prev = prompt.t().int().contiguous(), 0, None
        else:
            x, pos0, prev = seq[:, T + f - 1].view(1, ncb1).int().contiguous(), T + f - 1, window.clone()
        tok, slow, hidden, fast = step_fn(f, x, pos0, prev)
        tok = tok.long().cpu().view(-1)
        want = seq[:, T + f]
        e = ulp_err(slow, ref_slow[f])
        stats["max_ulp_slow"] = max(stats["max_ulp_slow"], e)
        assert e <= ulps, f"frame {f}: slow logits {e:.2f} ulps off the reference"
        e = ulp_err(hidden, ref_hidden[f])
        assert e <= ulps, f"frame {f}: hidden {e:.2f} ulps off the reference"
        # slow decision
        ref_tok = int(want[0])
        if decide == "equal":
            assert int(tok[0]) == ref_tok, f"frame {f}: slow token {int(tok[0])} != the reference's {ref_tok}"
        elif ref_tok == 0 and 0 not in ids.tolist():  # u == 0 quirk: independent of the logits
            assert int(tok[0]) == 0, f"frame {f}: the u==0 draw must return token 0"
        else:
            picked = (ids == int(tok[0])).nonzero()
            assert len(picked), f"frame {f}: token {int(tok[0])} is outside the constrained set"
            assert near_argmax(ref_slow[f], int(picked[0]), int((ids == ref_tok).nonzero()[0])), \
                f"frame {f}: slow token {int(tok[0])} is not a near-argmax of the reference logits"
        stats["decisions"] += 1
        stats["exact"] += int(int(tok[0]) == ref_tok)
        ok_chain = int(tok[0]) == ref_tok and int(tok[1]) == int(want[1])
        assert (int(tok[0]) != ref_tok) or int(tok[1]) == int(want[1]), f"frame {f}: codebook 0 mapping"
        for cb in range(1, cfg.num_codebooks):
            if not ok_chain:
                break  # a legitimately different code changes the rest of this frame's fast chain
            e = ulp_err(fast[cb - 1], ref_fast[f, cb - 1])
            stats["max_ulp_fast"] = max(stats["max_ulp_fast"], e)
            assert e <= ulps, f"frame {f} cb {cb}: fast logits {e:.2f} ulps off the reference"
            got_c, ref_c = int(tok[1 + cb]), int(want[1 + cb])
            ref_is_argmax = int(ref_fast[f, cb - 1].float().argmax()) == ref_c or \
                float(ref_fast[f, cb - 1].float()[ref_c]) == float(ref_fast[f, cb - 1].float().max())
            if decide == "equal":
                assert got_c == ref_c, f"frame {f} cb {cb}: code {got_c} != the reference's {ref_c}"
            elif not ref_is_argmax and ref_c == 0:  # u == 0 quirk
                assert got_c == 0, f"frame {f} cb {cb}: the u==0 draw must return code 0"
            else:
                assert near_argmax(ref_fast[f, cb - 1], got_c, ref_c), \
                    f"frame {f} cb {cb}: code {got_c} is not a near-argmax of the reference logits"
            stats["decisions"] += 1
            stats["exact"] += int(got_c == ref_c)
            ok_chain = got_c == ref_c
        if f > 0:
            window = window.roll(-1, dims=1)
            window[:, -1] = want.int()
        stats["frames"] += 1
    return stats


def oracle_step_fn(cfg, state, uniform_seed, temperature: float = 0.7, top_p: float = 0.7, top_k: int = 1, hook=None):
    """The CPU oracle behind the check_teacher_forced protocol.  `hook(orc)`: tests/mutations.py injects faults."""
    from oracle import dual_ar as O

    orc = O.DualAROracle(cfg, state)
    orc.setup_caches(1, cfg.max_seq_len)
    if hook is not None:
        hook(orc)
    bias = O.semantic_logit_bias(cfg, orc.dtype)
    temp = torch.tensor(temperature, dtype=orc.dtype)
    tp = torch.tensor(top_p, dtype=orc.dtype)
    u = O.FmiUniform(uniform_seed, 0)
    ncb1 = cfg.num_codebooks + 1

    def step(f, x, pos0, prev):
        orc.trace = {}
        u.frame, u.draw_idx = f, 0
        S = x.shape[0]
        xt = x.t().contiguous().view(1, ncb1, S).long()
        pos = torch.arange(pos0, pos0 + S)
        out = O.decode_one_token(orc, xt, pos, temp, tp, top_k, bias, prev, u, math_backend=(f > 0))
        live = torch.tensor(sorted(set(range(cfg.semantic_begin_id, cfg.semantic_end_id + 1)) | {cfg.im_end_id}))
        return (out.view(-1), orc.trace["slow_logits"][0][live], orc.trace["hidden"][0],
                torch.stack(orc.trace["fast_logits"][0]))

    return step

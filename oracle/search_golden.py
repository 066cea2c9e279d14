"""Seed search for WELL-CONDITIONED Dual-AR fixtures (test infrastructure; `python -m oracle.search_golden`).

Runs the CPU oracle (bit-identical to the reference on this machine, tests/test_oracle_cpu.py) over candidate
(state seed, prompt seed, uniform seed) triples of the peaky synthetic model (oracle.dual_ar.make_peaky_state) and
prints those whose free run is robust in EVERY decision:

  greedy (top_k = 1): top-1 margin of the constrained slow logits and of every fast logits row >= MIN_ULPS bf16 steps;
  sampled (top_k = 30): no decision changes under any tried logit perturbation of up to NOISE_ULPS bf16 steps
  (oracle.dual_ar.decision_noise_margin), and the run actually exercises the sampler (>= 1/8 of the slow
  tokens are not the top-1 candidate, RAS fires at least once).

The winners are copied into oracle/gen_golden.py's PEAKY_CASES, which re-runs them through the UNMODIFIED reference
and re-asserts the margins on the reference's own traces before writing tests/golden/.
"""
from __future__ import annotations

import sys

import torch

from . import dual_ar as O

MIN_ULPS = 10.0      # fixtures are required to have >= 8; search with slack
NOISE_ULPS = 2

MID = dict(vocab_size=2048, n_layer=3, n_head=4, n_local_heads=1, head_dim=128, dim=256, intermediate_size=512,
           codebook_size=256, num_codebooks=10, semantic_begin_id=1500, semantic_end_id=1755, im_end_id=1400,
           n_fast_layer=2, max_seq_len=512)


def live_ids(cfg):
    return torch.tensor(sorted(set(range(cfg.semantic_begin_id, cfg.semantic_end_id + 1)) | {cfg.im_end_id}))


def run(cfg, state, prompt, max_new, temperature, top_p, top_k, useed):
    orc = O.DualAROracle(cfg, state)
    orc.trace = {}
    y = O.generate(orc, prompt, max_new, temperature, top_p, top_k, uniform_fn=O.FmiUniform(useed, 0))
    return y, orc.trace


def greedy_min_margin(cfg, trace):
    ids = live_ids(cfg)
    slow = torch.stack(trace["slow_logits"])[:, ids]
    fast = torch.stack([torch.stack(f) for f in trace["fast_logits"]])
    return float(O.greedy_frame_margins(cfg, slow, fast).min())


def sampled_frame_is_robust(cfg, f, frame_tokens, slow_logits, fast_logits, window, temp, tp, top_k, useed, ulps, trials,
                            gen, bias):
    """One frame of a sampled run: re-derive its decisions from the traced logits and the generator's uniforms and test
    each under logit noise.  frame_tokens: (1+ncb,) the run's tokens of this frame; window: RAS window BEFORE the
    frame (None for the prefill frame).  Returns (robust, slow token is not the top-1 candidate, RAS fired)."""
    dt = slow_logits.dtype
    u = lambda d, n: (torch.from_numpy(O.fmi_uniform_u8(useed, 0, f, d, n).astype("float32")) / 256.0).to(dt)
    biased = slow_logits + bias
    w0 = window[0].clone() if window is not None else None
    u_n, u_h = u(0, cfg.vocab_size), u(1, cfg.vocab_size)
    dec = lambda lg: O.slow_decision(cfg, lg, temp, tp, top_k, u_n, u_h, w0)
    tok = dec(biased)
    assert tok == int(frame_tokens[0]), (f, tok, int(frame_tokens[0]))
    non_top1 = int(tok != int(biased.float().argmax()))
    ras = int(w0 is not None and bool((w0 == int(O.draw(O.logits_to_probs(biased, temp, tp, top_k), u_n))).any()))
    if not O.decision_noise_margin(dec, biased, ulps, trials, gen):
        return False, non_top1, ras
    for cb in range(1, cfg.num_codebooks):
        lg = fast_logits[cb - 1]
        u_c = u(1 + cb, cfg.codebook_size)
        decf = lambda l: int(O.draw(O.logits_to_probs(l, temp, tp, top_k), u_c))
        assert decf(lg) == int(frame_tokens[1 + cb]), (f, cb)
        if not O.decision_noise_margin(decf, lg, ulps, trials, gen):
            return False, non_top1, ras
    return True, non_top1, ras


def sampled_run_is_robust(cfg, y, trace, T, temperature, top_p, top_k, useed, ulps=NOISE_ULPS, trials=24):
    """Re-derive every decision of a sampled run from its traced logits and the generator's uniforms and test it
    under logit noise.  Returns (robust, n_non_top1, n_ras)."""
    dt = trace["slow_logits"][0].dtype
    temp = torch.tensor(temperature, dtype=dt)
    tp = torch.tensor(top_p, dtype=dt)
    bias = O.semantic_logit_bias(cfg, dt)[0, 0]
    gen = torch.Generator().manual_seed(1)
    ncb1 = 1 + cfg.num_codebooks
    window = torch.zeros((ncb1, O.RAS_WIN_SIZE), dtype=torch.int)
    non_top1 = ras = 0
    n_frames = y.shape[1] - T
    for f in range(n_frames):
        ok, nt, rs = sampled_frame_is_robust(cfg, f, y[:, T + f], trace["slow_logits"][f], trace["fast_logits"][f],
                                             window if f > 0 else None, temp, tp, top_k, useed, ulps, trials, gen, bias)
        non_top1 += nt
        ras += rs
        if not ok:
            return False, non_top1, ras
        if f > 0:
            window = window.roll(-1, dims=1)
            window[:, -1] = y[:, T + f].int()
    return True, non_top1, ras


def search_greedy(name, kw, gains, T, nsem, max_new, eos_after=None, seeds=range(1, 400)):
    cfg = O.DualARConfig(**kw)
    for sseed in seeds:
        state = O.make_peaky_state(cfg, sseed, **gains)
        for pseed in range(1, 4):
            prompt = O.make_prompt(cfg, T, seed=pseed, n_semantic=nsem)
            y, tr = run(cfg, state, prompt, max_new, 0.7, 0.7, 1, 1234)
            n = y.shape[1] - T
            if eos_after is None and n < max_new:
                continue
            if eos_after is not None and not (eos_after <= n < max_new):
                continue
            m = greedy_min_margin(cfg, tr)
            if m >= MIN_ULPS and len(set(y[0, T:].tolist())) >= n // 2:
                print(f"GREEDY {name}: state_seed={sseed} prompt_seed={pseed} frames={n} min_margin={m:.1f} "
                      f"distinct_slow={len(set(y[0, T:].tolist()))}", flush=True)
                return sseed, pseed
    return None


def search_sampled(name, kw, gains, T, nsem, max_new, temperature, top_p, top_k, seeds=range(1, 60), useeds=range(1, 400)):
    cfg = O.DualARConfig(**kw)
    for sseed in seeds:
        state = O.make_peaky_state(cfg, sseed, **gains)
        prompt = O.make_prompt(cfg, T, seed=sseed, n_semantic=nsem)
        for useed in useeds:
            y, tr = run(cfg, state, prompt, max_new, temperature, top_p, top_k, useed)
            if y.shape[1] - T < max_new:
                continue
            ok, nt, ras = sampled_run_is_robust(cfg, y, tr, T, temperature, top_p, top_k, useed)
            if ok and nt >= max_new // 8 and ras >= 1:
                print(f"SAMPLED {name}: state_seed={sseed} uniform_seed={useed} non_top1={nt} ras={ras}", flush=True)
                return sseed, useed
    return None


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["tiny", "mid", "tiny_eos", "tiny_sampled"]
    G = dict(emb_gain=1.5, slow_gain=2.0, fast_gain=1.5)
    if "tiny" in which:
        search_greedy("tiny_peaky", {}, G, 24, 8, 64)
    if "tiny_eos" in which:
        search_greedy("tiny_peaky_eos", {}, dict(G, eos_code=17), 24, 8, 96, eos_after=40)
    if "mid" in which:
        search_greedy("mid_peaky", MID, dict(emb_gain=2.5, slow_gain=3.0, fast_gain=2.5), 40, 12, 48)
    if "tiny_sampled" in which:
        search_sampled("tiny_sampled", {}, dict(emb_gain=6.0, slow_gain=2.0, fast_gain=8.0, hot=(1.0, 0.95, 0.93), hot_every=3),
                       24, 8, 40, 0.7, 0.9, 30)

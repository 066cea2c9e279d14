"""Golden fixtures at the BASELINE model width: the UNMODIFIED reference's generate() on the S2-Pro-shaped 4.56 B
parameter model (authoring container only: needs /root/reference, ~25 GB of RAM and a few minutes per case).
`python -m oracle.gen_golden_s2 [case ...]`.

Weights: `oracle.dual_ar.make_peaky_state_hash` -- every tensor a pure integer-arithmetic function of (seed, tensor
name, element index), so the GPU box re-creates the very same bf16 model on the device in seconds (torch's CPU and GPU
generators differ, which is why the earlier full-width tests could only compare against an oracle run on the box).
Gains: at 36 layers the residual stream is dominated by the layers' own outputs (rms ~0.6 per layer against 0.03 for
an embedding row), so the codebook-0 row that points at the successor needs slow_gain ~200 for the successor's logit to
stand ~10 sigma above the other 4096 (measured: slow_gain 100 -> top logit 8 against a noise maximum of 6; 300 -> 24),
and the fast embeddings need x30.

What is written (tests/golden/dualar_s2_*.npz, a few KB each): the prompt, the token matrix generate() returned, the
per-frame minimum decision margin measured on the reference's own logits, the sampling parameters and the weight
recipe.  Asserted before writing: the oracle reproduces the reference run bit for bit on this machine; greedy cases
have >= MIN_MARGIN bf16 steps at every decision; the sampled case is invariant under NOISE_ULPS steps of logit noise
at every decision, draws non-top-1 tokens and fires RAS."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from . import dual_ar as O
from .gen_golden import OUT, _ref_generate, live_ids
from .refload import FakeTokenizer, add_reference_to_path
from . import refload

MIN_MARGIN = 16.0
NOISE_ULPS = 2

GREEDY_STATE = dict(seed=1, emb_gain=1.5, slow_gain=200.0, fast_gain=1.0, fast_emb_gain=30.0)
SAMPLED_STATE = dict(seed=2, emb_gain=1.5, slow_gain=200.0, fast_gain=1.0, fast_emb_gain=30.0, hot=(1.0, 0.95, 0.93),
                     hot_every=8, pair_cycles=True)
# round 4 (VERDICT r03 weak #1a): flatter, more frequent extra successors so that robust sampled decisions LEAVE the
# top-1 candidate at the BASELINE width too (the round-3 recipe above fired RAS 61x but never did)
SAMPLED2_STATE = dict(seed=2, emb_gain=1.5, slow_gain=200.0, fast_gain=1.0, fast_emb_gain=30.0,
                      hot=tuple(float(v) for v in os.environ.get("S2_HOT", "1.0,0.985,0.97").split(",")),
                      hot_every=int(os.environ.get("S2_HOT_EVERY", "2")), pair_cycles=True)

CASES = {
    # name: (state kwargs, prompt (T, n_semantic, candidate seeds), frames, (temperature, top_p, top_k), candidate uniform seeds)
    "s2_plain": (GREEDY_STATE, (200, 0, range(1, 40)), 64, (0.7, 0.7, 1), range(1234, 1260)),
    "s2_clone": (GREEDY_STATE, (200, 100, range(1, 8)), 64, (0.7, 0.7, 1), range(1234, 1260)),
    # uniform seed 6 is the first whose 64 frames are all robust (12 others broke off at a fragile decision); its run
    # fires RAS in 61 frames (the high-temperature draw replaces the normal one) but never leaves the top-1 candidate
    "s2_sampled": (SAMPLED_STATE, (200, 60, range(1, 8)), 64, (0.7, 0.9, 30), range(int(os.environ.get("S2_USEED0", "6")), 200)),
    # round 4: the benchmark's full 215 frames (context 200 -> 415: all 7 KV pages of a 512-position slot), greedy
    "s2_plain215": (GREEDY_STATE, (200, 0, range(1, 40)), 215, (0.7, 0.7, 1), range(1234, 1260)),
    # round 4: a 1010-token voice-clone-shaped prompt + 24 frames, greedy: the free run crosses position 1024, where decode
    # attention moves to the MFMA split-K kernel + merge -- at the BASELINE width (G = 4, D = 128: the kernels the
    # benchmark shape runs at long contexts), with a 1010-row prefill through the 256-column GEMM tiles
    "s2_long": (GREEDY_STATE, (1010, 300, range(1, 12)), 24, (0.7, 0.7, 1), range(1234, 1240)),
    # round 4: sampled decisions that leave the top-1 candidate (S2_MIN_NON_TOP1, default 3 for this case)
    "s2_sampled2": (SAMPLED2_STATE, (200, 60, range(1, 8)), 64, (0.7, 0.9, 30), range(int(os.environ.get("S2_USEED0", "1")), 400)),
}
# round 4: the six OTHER rows of tests/test_s2_parity_gpu.py's ragged batch of 8 (rows 2 and 5 are s2_plain / s2_clone),
# so that every row of the batch is the unmodified reference's: (row, T, n_semantic, candidate prompt seeds)
RAGGED_ROWS = [(0, 57, 0, range(300, 340)), (1, 333, 111, range(301, 340)), (3, 131, 43, range(303, 340)),
               (4, 64, 0, range(304, 340)), (6, 400, 0, range(306, 340)), (7, 90, 30, range(307, 340))]


def build_reference_meta(cfg, state):
    """The reference DualARTransformer over the given tensors without the 18 GB fp32 random init of its constructor:
    built on the meta device, parameters assigned from `state`, the two non-persistent buffers rebuilt by the
    reference's own functions."""
    add_reference_to_path()
    from fish_speech.models.text2semantic import llama as RL

    args = RL.DualARModelArgs(**cfg.reference_kwargs())
    with torch.device("meta"):
        model = RL.DualARTransformer(args)
    missing, unexpected = model.load_state_dict(state, strict=False, assign=True)
    assert not unexpected, unexpected
    assert all(m in ("freqs_cis", "causal_mask", "fast_freqs_cis") for m in missing), missing
    model.register_buffer("freqs_cis", RL.precompute_freqs_cis(args.max_seq_len, args.head_dim, args.rope_base), persistent=False)
    model.register_buffer("causal_mask", torch.tril(torch.ones(args.max_seq_len, args.max_seq_len, dtype=torch.bool)), persistent=False)
    model.register_buffer("fast_freqs_cis", RL.precompute_freqs_cis(args.num_codebooks, args.fast_head_dim, args.rope_base), persistent=False)
    for n, p in model.named_parameters():
        assert p.device.type == "cpu" and p.dtype == torch.bfloat16, n
    model = model.eval()
    model.tokenizer = FakeTokenizer(cfg.im_end_id)
    model._cache_setup_done = False
    return model


class _NotRobust(Exception):
    pass


class _CheckedUniform(O.FmiUniform):
    """FmiUniform that, whenever generate() moves to the next frame, tests the frame just finished for robustness and
    aborts the run (an attempt costs the frames up to its first fragile decision, not all 64)."""

    def __init__(self, seed, check):
        super().__init__(seed, 0)
        self.check = check

    def next_frame(self):
        self.check(self.frame)
        super().next_frame()


def main(which):
    from .search_golden import sampled_frame_is_robust, sampled_run_is_robust

    torch.set_num_threads(8)
    cfg = O.s2_pro_shaped_config(max_seq_len=2048 if "s2_long" in which else 512)
    ids = live_ids(cfg)
    states = {}
    orig_build = refload.build_reference_dual_ar
    import oracle.gen_golden as GG

    for name in which:
        skw, (T, nsem, pseeds), frames, (temp, top_p, top_k), useeds = CASES[name]
        key = json.dumps(skw, sort_keys=True)
        if key not in states:
            states.clear()
            t0 = time.time()
            states[key] = O.make_peaky_state_hash(cfg, **skw)
            print(f"{name}: state built in {time.time() - t0:.0f}s", flush=True)
        state = states[key]
        orc = O.DualAROracle(cfg, state)
        found = None
        # the ORACLE (0.5 s/frame) searches the prompt / uniform seeds; the reference then re-runs the winner
        for pseed in pseeds:
            prompt = O.make_prompt(cfg, T, seed=pseed, n_semantic=nsem)
            for useed in useeds:
                orc.trace = {}
                t0 = time.time()
                ufn = O.FmiUniform(useed, 0)
                if top_k != 1:   # sampled: test every frame as soon as it exists
                    dt = torch.bfloat16
                    st_ = dict(window=torch.zeros((1 + cfg.num_codebooks, O.RAS_WIN_SIZE), dtype=torch.int),
                               gen=torch.Generator().manual_seed(1))
                    bias_ = O.semantic_logit_bias(cfg, dt)[0, 0]

                    def check(f, useed=useed, st_=st_):
                        tr_ = orc.trace
                        toks = torch.cat([torch.tensor([tr_["slow_token"][f]]),
                                          torch.tensor([max(0, min(tr_["slow_token"][f] - cfg.semantic_begin_id, cfg.codebook_size - 1))]),
                                          torch.tensor([int(O.draw(O.logits_to_probs(l, torch.tensor(temp, dtype=dt), torch.tensor(top_p, dtype=dt), top_k),
                                                                   (torch.from_numpy(O.fmi_uniform_u8(useed, 0, f, 1 + cb, cfg.codebook_size).astype("float32")) / 256.0).to(dt)))
                                                        for cb, l in enumerate(tr_["fast_logits"][f], start=1)])])
                        ok_, _, _ = sampled_frame_is_robust(cfg, f, toks, tr_["slow_logits"][f], tr_["fast_logits"][f],
                                                            st_["window"] if f > 0 else None, torch.tensor(temp, dtype=dt),
                                                            torch.tensor(top_p, dtype=dt), top_k, useed, NOISE_ULPS, 24,
                                                            st_["gen"], bias_)
                        if not ok_ or toks[0] == 0:     # (a u == 0 draw leaves the peaky chain: the next frame is a coin flip)
                            raise _NotRobust(f)
                        if f > 0:
                            st_["window"] = st_["window"].roll(-1, dims=1)
                            st_["window"][:, -1] = toks.int()

                    ufn = _CheckedUniform(useed, check)
                try:
                    y = O.generate(orc, prompt, frames, temp, top_p, top_k, uniform_fn=ufn, stop_on_im_end=False)
                except _NotRobust as e:
                    print(f"  {name}: prompt seed {pseed} uniform seed {useed}: fragile decision in frame {e.args[0]} "
                          f"({time.time() - t0:.0f}s)", flush=True)
                    continue
                slow_full = torch.stack(orc.trace["slow_logits"])
                fast = torch.stack([torch.stack(f) for f in orc.trace["fast_logits"]])
                margins = O.greedy_frame_margins(cfg, slow_full[:, ids], fast)
                n0 = int((y[0, T:] == 0).sum())
                if top_k == 1:
                    ok = float(margins.min()) >= MIN_MARGIN
                    note = f"min margin {float(margins.min()):.1f}, u==0 slow tokens {n0}"
                else:
                    rtr = {"slow_logits": list(slow_full), "fast_logits": [list(f) for f in fast]}
                    rob, nt, ras = sampled_run_is_robust(cfg, y, rtr, T, temp, top_p, top_k, useed, ulps=NOISE_ULPS)
                    ok = rob and nt >= int(os.environ.get("S2_MIN_NON_TOP1", "3" if name == "s2_sampled2" else "0")) and ras >= 8
                    note = f"robust {rob}, non-top-1 {nt}, RAS {ras}, u==0 slow tokens {n0}"
                print(f"  {name}: prompt seed {pseed} uniform seed {useed}: {note} ({time.time() - t0:.0f}s)", flush=True)
                if ok:
                    found = (pseed, useed, prompt, y, margins, note)
                    break
                if top_k == 1 and float(margins[0]) < MIN_MARGIN:
                    break   # the first decision depends on the prompt only: next prompt
            if found:
                break
        assert found, name
        pseed, useed, prompt, y, margins, note = found
        # the UNMODIFIED reference on the same tensors
        GG.build_reference_dual_ar = build_reference_meta
        try:
            t0 = time.time()
            tr = {}
            tokens = _ref_generate(cfg, state, prompt, frames, top_k, O.FmiUniform(seed=useed, stream=0), trace=tr,
                                   temperature=temp, top_p=top_p)
            print(f"  {name}: reference generate() {time.time() - t0:.0f}s", flush=True)
        finally:
            GG.build_reference_dual_ar = orig_build
        # generate() stops at <|im_end|>; the cases are chosen not to contain one
        assert tokens.shape == y.shape and torch.equal(tokens, y), f"{name}: oracle != reference"
        slow_ref = torch.stack(tr["slow_logits"])[:, ids]
        fast_ref = torch.stack([torch.stack(f) for f in tr["fast_logits"]])
        m_ref = O.greedy_frame_margins(cfg, slow_ref, fast_ref)
        assert torch.equal(m_ref, margins), "the reference's own logits give other margins than the oracle's"
        np.savez_compressed(
            os.path.join(OUT, f"dualar_{name}.npz"), prompt=prompt.numpy(), tokens=tokens.numpy(),
            state_kind="peaky_hash", state_kwargs=json.dumps(skw), max_new=frames, uniform_seed=useed,
            prompt_seed=pseed, temperature=temp, top_p=top_p, top_k=top_k, greedy_margins_ulps=margins.numpy(),
            note=note)
        print(f"{name}: written; {note}", flush=True)


def _reference_run(cfg, state, prompt, frames, top_k, useed, temp, top_p, int8=False):
    import oracle.gen_golden as GG

    orig_build = refload.build_reference_dual_ar
    GG.build_reference_dual_ar = build_reference_meta
    try:
        tr = {}
        tokens = _ref_generate(cfg, state, prompt, frames, top_k, O.FmiUniform(seed=useed, stream=0), trace=tr,
                               temperature=temp, top_p=top_p, int8=int8)
    finally:
        GG.build_reference_dual_ar = orig_build
    return tokens, tr


def main_ragged():
    """Six more greedy utterances of ragged prompt lengths on the GREEDY_STATE weights (one fixture file): searched
    with the oracle for >= MIN_MARGIN at every decision, then re-run through the unmodified reference."""
    torch.set_num_threads(8)
    cfg = O.s2_pro_shaped_config(max_seq_len=512)
    ids = live_ids(cfg)
    t0 = time.time()
    state = O.make_peaky_state_hash(cfg, **GREEDY_STATE)
    print(f"s2_ragged: state built in {time.time() - t0:.0f}s", flush=True)
    orc = O.DualAROracle(cfg, state)
    out = {}
    for row, T, nsem, pseeds in RAGGED_ROWS:
        found = None
        for pseed in pseeds:
            prompt = O.make_prompt(cfg, T, seed=pseed, n_semantic=nsem)
            orc.trace = {}
            t0 = time.time()
            y = O.generate(orc, prompt, 64, 0.7, 0.7, 1, uniform_fn=O.FmiUniform(700 + row, 0), stop_on_im_end=False)
            slow = torch.stack(orc.trace["slow_logits"])[:, ids]
            fast = torch.stack([torch.stack(f) for f in orc.trace["fast_logits"]])
            margins = O.greedy_frame_margins(cfg, slow, fast)
            print(f"  row {row} T={T}: prompt seed {pseed}: min margin {float(margins.min()):.1f} ({time.time() - t0:.0f}s)", flush=True)
            if float(margins.min()) >= MIN_MARGIN:
                found = (pseed, prompt, y, margins)
                break
        assert found, row
        pseed, prompt, y, margins = found
        t0 = time.time()
        tokens, tr = _reference_run(cfg, state, prompt, 64, 1, 700 + row, 0.7, 0.7)
        print(f"  row {row}: reference generate() {time.time() - t0:.0f}s", flush=True)
        assert tokens.shape == y.shape and torch.equal(tokens, y), f"row {row}: oracle != reference"
        out[f"prompt_{row}"] = prompt.numpy()
        out[f"tokens_{row}"] = tokens.numpy()
        out[f"margins_{row}"] = margins.numpy()
        out[f"prompt_seed_{row}"] = pseed
    np.savez_compressed(os.path.join(OUT, "dualar_s2_ragged.npz"), rows=np.array([r[0] for r in RAGGED_ROWS]),
                        state_kind="peaky_hash", state_kwargs=json.dumps(GREEDY_STATE), max_new=64,
                        uniform_seed_base=700, temperature=0.7, top_p=0.7, top_k=1, **out)
    print("s2_ragged: written", flush=True)


def main_int8():
    """The s2_plain utterance on the SAME weights quantised by the reference's own WeightOnlyInt8QuantHandler
    (tools/llama/quantize.py:186-229), generated by the unmodified reference through its int8 Linear
    (llama.py:529-534): the S2-width int8 fixture bench.py --int8's number rests on (VERDICT r03 weak #1c)."""
    torch.set_num_threads(8)
    cfg = O.s2_pro_shaped_config(max_seq_len=512)
    ids = live_ids(cfg)
    state = O.make_peaky_state_hash(cfg, **GREEDY_STATE)
    z = np.load(os.path.join(OUT, "dualar_s2_plain.npz"))
    prompt = torch.from_numpy(z["prompt"])
    frames = int(os.environ.get("S2_INT8_FRAMES", "64"))
    t0 = time.time()
    tokens, tr = _reference_run(cfg, state, prompt, frames, 1, int(z["uniform_seed"]), 0.7, 0.7, int8=True)
    print(f"s2_int8: reference int8 generate() {time.time() - t0:.0f}s", flush=True)
    slow = torch.stack(tr["slow_logits"])[:, ids]
    fast = torch.stack([torch.stack(f) for f in tr["fast_logits"]])
    margins = O.greedy_frame_margins(cfg, slow, fast)
    print(f"s2_int8: min margin {float(margins.min()):.1f}", flush=True)
    assert float(margins.min()) >= MIN_MARGIN, float(margins.min())
    t0 = time.time()
    orc = O.DualAROracle(cfg, O.quantize_state_int8(cfg, state))
    mine = O.generate(orc, prompt, frames, 0.7, 0.7, 1, uniform_fn=O.FmiUniform(int(z["uniform_seed"]), 0), stop_on_im_end=False)
    print(f"s2_int8: oracle int8 generate() {time.time() - t0:.0f}s", flush=True)
    assert torch.equal(mine, tokens), "s2_int8: oracle != reference"
    np.savez_compressed(os.path.join(OUT, "dualar_s2_int8.npz"), prompt=prompt.numpy(), tokens=tokens.numpy(),
                        state_kind="peaky_hash", state_kwargs=json.dumps(GREEDY_STATE), max_new=frames,
                        uniform_seed=int(z["uniform_seed"]), prompt_seed=int(z["prompt_seed"]), temperature=0.7, top_p=0.7,
                        top_k=1, greedy_margins_ulps=margins.numpy(), same_tokens_as_bf16=bool(np.array_equal(tokens.numpy(), z["tokens"])),
                        note=f"int8 weights by the reference's WeightOnlyInt8QuantHandler; min margin {float(margins.min()):.1f}")
    print("s2_int8: written", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:] or list(CASES)
    if "s2_ragged" in args:
        main_ragged()
    if "s2_int8" in args:
        main_int8()
    rest = [a for a in args if a in CASES]
    if rest:
        main(rest)

"""Index parity at the BASELINE model width (S2-Pro shape, 4.56 B parameters, 36 + 4 layers), ASSERTED AS EQUALITY.

tests/golden/dualar_s2_*.npz were written by the UNMODIFIED reference's generate() (inference.py:243-359) run on the
authoring container's CPU over a well-conditioned ("peaky", oracle.dual_ar.make_peaky_state_hash) full-width model:
200-token prompts (one plain text, one voice-clone shaped), 64 free-running frames, greedy and sampled (top-k 30,
top-p 0.9, temperature 0.7, RAS firing).  The weights are a pure integer-arithmetic function of (seed, tensor name,
element index), so this box re-creates the very same bf16 tensors on the GPU in seconds.  Every decision of the greedy
runs has >= 16 bf16 steps of margin on the reference's own logits (measured 39 and 57), the sampled run is invariant
under 4 steps of logit noise at every decision (oracle/gen_golden_s2.py asserts both before writing), so any correct
implementation must return the SAME token matrix: 64 x (1 slow + 9 fast) decisions per case through the S2-only code
paths -- the balanced decode GEMVs, D = 128 / G = 4 attention over 4+ KV pages, the live-row LM head, the per-code
q|k|v table of fast layer 0."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dual_ar as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# s2_plain215 (round 4): the benchmark's full 215 frames (context 200 -> 415, all 7 KV pages of a 512-position slot);
# s2_sampled2 (round 4): sampled decisions that leave the top-1 candidate at the BASELINE width
CASES = [c for c in ("s2_plain", "s2_clone", "s2_sampled", "s2_plain215", "s2_sampled2")
         if os.path.exists(os.path.join(GOLDEN, f"dualar_{c}.npz"))]


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"dualar_{name}.npz"))
    skw = json.loads(str(z["state_kwargs"]))
    if "hot" in skw:
        skw["hot"] = tuple(skw["hot"])
    return z, skw


_models = {}


def _model(skw, max_seq_len=512, slots=16):
    """One HIP model per weight recipe (greedy cases share theirs), built on the device; kept for the module."""
    from fish_speech_amd.dual_ar import MiDualAR

    key = json.dumps(skw, sort_keys=True) + f"|{max_seq_len}"
    if key not in _models:
        ocfg = O.s2_pro_shaped_config(max_seq_len=max_seq_len)
        state = O.make_peaky_state_hash(ocfg, device=DEV, **skw)
        m = MiDualAR(ocfg, device=DEV, im_end_id=ocfg.im_end_id)
        m.load_state_dict(state)
        m.setup_caches(slots, max_seq_len)
        _models[key] = (ocfg, m, state)
    return _models[key]


def test_hash_weights_are_bit_identical_on_cpu_and_gpu():
    """The premise of these fixtures: the counter-based weight generator gives the same bits on both devices."""
    for shape, key, std, mean in [((4096, 2560), 77, 0.02, 0.0), ((2560,), 5, 0.1, 1.0), ((1000, 33), 123456789, 0.02, 0.0)]:
        a = O.hash_normal(shape, key, std, "cpu", mean=mean)
        b = O.hash_normal(shape, key, std, DEV, mean=mean).cpu()
        assert torch.equal(a, b) and torch.equal(a.bfloat16(), b.bfloat16())
    assert abs(float(O.hash_normal((1 << 20,), 9, 0.02).std()) - 0.02) < 2e-4


@pytest.mark.parametrize("case", CASES)
def test_s2_full_sequence_equals_the_reference(case):
    """prefill (tiled GEMM + MFMA flash attention over 200 positions) + 63 hipGraph-replayed frames, then the eager
    path with EOS polled every frame: the whole (11, 264) token matrix equals what the reference's generate() returned."""
    from fish_speech_amd.dual_ar import generate

    z, skw = _load(case)
    cfg, model, _ = _model(skw)
    want = z["tokens"]
    kw = dict(prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]), temperature=float(z["temperature"]),
              top_p=float(z["top_p"]), top_k=int(z["top_k"]), seed=int(z["uniform_seed"]))
    assert z["prompt"].shape[1] == 200 and want.shape[1] - 200 >= 64
    model.set_graph(True)
    model.set_fast_merge(True)
    got = generate(model=model, **kw).numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"{case}: first mismatch at (row, column) {bad[0].tolist()}: got {got[tuple(bad[0])]}, want {want[tuple(bad[0])]}"
    model.set_graph(False)
    got2 = generate(model=model, poll_every=1, **kw).numpy()
    model.set_graph(True)
    assert np.array_equal(got2, want), f"{case}: eager path differs"
    # fast positions 0 and 1 as two passes over the fast weights (rounds 1-3) instead of one: the same tokens
    model.set_fast_merge(False)
    got3 = generate(model=model, **kw).numpy()
    model.set_fast_merge(True)
    assert np.array_equal(got3, want), f"{case}: two-pass fast positions 0 / 1 differ"
    print(case, "full sequence equal;", str(z["note"]))


def test_s2_golden_utterances_inside_a_ragged_batch_of_8():
    """A ragged batch of 8 (config 4's mixed lengths: batch-8 GEMV variants, eight slots' pages interleaved in the
    pool) in which EVERY row is the unmodified reference's: rows 2 and 5 are s2_plain / s2_clone, the other six come
    from dualar_s2_ragged.npz (round 4; oracle/gen_golden_s2.py s2_ragged: prompts of 57 / 333 / 131 / 64 / 400 / 90
    tokens, every decision >= 16 bf16 steps of margin).  All eight token matrices must be equal."""
    from fish_speech_amd.dual_ar import generate_batch

    zp, skw = _load("s2_plain")
    zc, _ = _load("s2_clone")
    zr, skw_r = _load("s2_ragged")
    assert skw_r == skw
    cfg, model, _ = _model(skw)
    lens = [57, 333, 200, 131, 64, 200, 400, 90]
    prompts, seeds, want = [None] * 8, [None] * 8, [None] * 8
    prompts[2], seeds[2], want[2] = torch.from_numpy(zp["prompt"]), int(zp["uniform_seed"]), zp["tokens"]
    prompts[5], seeds[5], want[5] = torch.from_numpy(zc["prompt"]), int(zc["uniform_seed"]), zc["tokens"]
    for row in zr["rows"].tolist():
        prompts[row], seeds[row] = torch.from_numpy(zr[f"prompt_{row}"]), int(zr["uniform_seed_base"]) + row
        want[row] = zr[f"tokens_{row}"]
    assert [p.shape[1] for p in prompts] == lens
    out = generate_batch(model=model, prompts=prompts, max_new_tokens=64, temperature=0.7, top_p=0.7, top_k=1,
                         seeds=seeds, stop_on_im_end=False)
    for row in range(8):
        assert np.array_equal(out[row].numpy(), want[row]), f"row {row} (T = {lens[row]}) differs from the reference"


def _ragged_prompts():
    zp, skw = _load("s2_plain")
    zc, _ = _load("s2_clone")
    zr, _ = _load("s2_ragged")
    prompts, seeds = [None] * 8, [None] * 8
    prompts[2], seeds[2] = torch.from_numpy(zp["prompt"]), int(zp["uniform_seed"])
    prompts[5], seeds[5] = torch.from_numpy(zc["prompt"]), int(zc["uniform_seed"])
    for row in zr["rows"].tolist():
        prompts[row], seeds[row] = torch.from_numpy(zr[f"prompt_{row}"]), int(zr["uniform_seed_base"]) + row
    return skw, prompts, seeds


def _run_with_taps(model, prompts, seeds, frames, trace=True):
    """prefill + `frames` graph-replayed frames of a batch; -> (token matrices, slow logits, hidden, fast logits of the
    last frame), the float taps as raw bf16 bit patterns.  trace=True: the logits of every fast position (the trace
    switches the per-code q|k|v table of fast layer 0 off, so layer 0 runs its GEMV); trace=False: the last position's
    logits only, with the table in use -- the frame the benchmark runs."""
    n = len(prompts)
    slots = list(range(n))
    samp = [model._sampling(0.7, 0.7, 1, seeds[i], True) for i in range(n)]
    model.set_trace(trace)
    model.set_graph(not trace)        # (the traced frames run eagerly; graph == eager is asserted by the sequence tests)
    model.prefill(slots, prompts, [frames + 1] * n, samp)
    model.decode(slots, frames)
    model.synchronize()
    logits, _, hidden, last = model.debug_taps(n)
    fast = model.fast_trace(n) if trace else last
    toks = [model.read(i)[0].clone() for i in slots]
    for i in slots:
        model.release(i)
    model.set_trace(False)
    model.set_graph(True)
    bits = lambda t: t.contiguous().view(torch.int16).cpu()   # noqa: E731
    return toks, bits(logits), bits(hidden), bits(fast)


@pytest.mark.parametrize("B", [5, 8, 12, 16])
def test_merged_fast_positions_equal_the_two_pass_path_bit_for_bit_at_batch_5_and_8(B):
    """with 5..8 utterances the merged pass runs the fast GEMVs at M = 2 B = 10..16 rows -- the 16-row
    forms, whose SwiGLU variant takes its RMSNorm statistics in another kernel branch than the <= 8-row form of the two-pass
    path.  Both branches now share one partition and reduction tree; asserted on the FLOAT taps, not only on tokens:
    slow logits, hidden rows and the logits of all ten fast positions of the last frame are bit-identical with the
    merge on and off, as are the token matrices (and the rebuilt key / value 0 of fast_attn_kernel's merged form with
    them: any difference there would move the position-1 logits).
    Round 6: B = 12 / 16 -- the merged pass at M = 24 / 32 rows (the GEMV's two-column-set form, XR = 32) against the
    two-pass path at M = 12 / 16 (the native 16-row form)."""
    skw, prompts, seeds = _ragged_prompts()
    prompts, seeds = prompts + prompts, seeds + seeds
    cfg, model, _ = _model(skw)
    for trace in (True, False):
        outs = {}
        for merge in (True, False):
            model.set_fast_merge(merge)
            outs[merge] = _run_with_taps(model, prompts[:B], seeds[:B], frames=6 if B <= 8 else 3, trace=trace)
        model.set_fast_merge(True)
        (t1, l1, h1, f1), (t0, l0, h0, f0) = outs[True], outs[False]
        for a, b in zip(t1, t0):
            assert torch.equal(a, b)
        assert torch.equal(l1, l0) and torch.equal(h1, h0), "slow taps differ between the merged and the two-pass frame"
        assert torch.equal(f1, f0), f"trace={trace}: fast logits differ at {torch.nonzero(f1 != f0)[:4].tolist()}"
        assert int(f1.ne(0).sum()) > 0


def test_rows_are_bit_identical_in_a_batch_of_8_and_in_a_batch_of_16():
    """Batch invariance across the M = 8 / M = 16 / M = 32 kernel forms, on the float taps: the eight ragged
    utterances alone, and twice over in a batch of 16 (slots i and i + 8 carry the same utterance) -- slow layers at
    M = 16 (the native 16-row form, round 6; SwiGLU without held fragments), fast positions 0/1 merged at M = 32 (two
    column sets) or, merge off, two passes at M = 16.  Every row's slow logits, hidden state and fast logits equal its
    batch-8 bits, with the table in use (trace off: the benchmark's frame) and without."""
    skw, prompts, seeds = _ragged_prompts()
    cfg, model, _ = _model(skw)
    for trace in (True, False):
        t8, l8, h8, f8 = _run_with_taps(model, prompts, seeds, frames=4, trace=trace)
        for merge in (True, False):
            model.set_fast_merge(merge)
            try:
                t16, l16, h16, f16 = _run_with_taps(model, prompts + prompts, seeds + seeds, frames=4, trace=trace)
            finally:
                model.set_fast_merge(True)
            for i in range(8):
                assert torch.equal(t16[i], t8[i]) and torch.equal(t16[i + 8], t8[i]), i
            for half in (slice(0, 8), slice(8, 16)):
                assert torch.equal(l16[half], l8) and torch.equal(h16[half], h8), "slow taps depend on the batch size"
                assert torch.equal(f16[half], f8), f"fast logits depend on the batch size (merge={merge}, trace={trace})"


def test_s2_int8_full_sequence_equals_the_reference_int8_run():
    """The S2-width int8 fixture (round 4; what `bench.py --int8` rests on): the s2_plain utterance on the same hash
    weights quantised by the reference's own WeightOnlyInt8QuantHandler (tools/llama/quantize.py:186-229) and generated
    by the unmodified reference through its int8 Linear (llama.py:529-534).  The HIP path loaded from the int8
    checkpoint (int8 tiles streamed by the decode GEMVs, scales in the epilogues) returns the same (11, 264) matrix."""
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate

    z, skw = _load("s2_int8")
    ocfg = O.s2_pro_shaped_config(max_seq_len=512)
    state = O.make_peaky_state_hash(ocfg, device=DEV, **skw)
    q = O.quantize_state_int8(ocfg, state)
    del state
    mcfg = DualARConfig.from_any(ocfg)
    mcfg.weight_int8 = True
    model = MiDualAR(mcfg, device=DEV, im_end_id=ocfg.im_end_id).load_state_dict(q)
    del q
    model.setup_caches(2, 512)
    got = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]), temperature=0.7,
                   top_p=0.7, top_k=1, seed=int(z["uniform_seed"])).numpy()
    want = z["tokens"]
    bad = np.argwhere(got != want) if got.shape == want.shape else None
    assert got.shape == want.shape and len(bad) == 0, f"first mismatch at {None if bad is None else bad[0].tolist()}"
    print("s2_int8 full sequence equal;", str(z["note"]))


def test_s2_oracle_on_this_box_equals_the_fixture_and_bounds_the_taps():
    """Ties the three implementations together on this machine: the CPU oracle, run here on the tensors copied back
    from the GPU, free-runs the voice-clone case to the SAME tokens the reference wrote in the authoring container
    (another CPU, another thread count), and the HIP path, teacher-forced through the decode_one_token seam, stays
    within the calibrated bf16 noise of the oracle's logits / hidden taps at every frame (relative L2 <= 8 %, the
    bound of test_s2_shape_forward_passes_match_the_cpu_oracle) with every one of the 640 decisions equal."""
    from tests.helpers import check_teacher_forced
    from tests.test_dualar_gpu import hip_step_fn

    z, skw = _load("s2_clone")
    cfg, model, state = _model(skw)
    host = {k: v.cpu() for k, v in state.items()}
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(nthreads, 16))
    try:
        orc = O.DualAROracle(cfg, host)
        orc.trace = {}
        seq = O.generate(orc, torch.from_numpy(z["prompt"]), int(z["max_new"]), float(z["temperature"]),
                         float(z["top_p"]), int(z["top_k"]), uniform_fn=O.FmiUniform(int(z["uniform_seed"]), 0))
    finally:
        torch.set_num_threads(nthreads)
    assert np.array_equal(seq.numpy(), z["tokens"]), "the oracle on this box disagrees with the reference's fixture"
    ids = model._table(1, torch.int32).view(-1).long().cpu()
    u16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)   # noqa: E731
    slow = torch.stack(orc.trace["slow_logits"])[:, ids]
    fast = torch.stack([torch.stack(f) for f in orc.trace["fast_logits"]])
    margins = O.greedy_frame_margins(cfg, slow, fast)
    assert float(margins.min()) >= 16.0, float(margins.min())

    class Z(dict):
        files = ["tokens"]

    zz = Z(tokens=seq.numpy(), prompt=z["prompt"], live_ids=ids.numpy(), slow_logits_live=u16(slow),
           hidden=u16(torch.stack(orc.trace["hidden"])), fast_logits=u16(fast))
    st = check_teacher_forced(hip_step_fn(model, cfg, int(z["uniform_seed"])), cfg, zz, ulps=1e9, decide_ulps=0.0,
                              rel_l2=0.08)
    model.set_trace(False)
    print("S2 peaky teacher-forced:", st, "min margin on this box", float(margins.min()))
    n = seq.shape[1] - z["prompt"].shape[1]
    assert st["frames"] == n and st["exact"] == st["decisions"] == n * cfg.num_codebooks, st


def test_s2_codes_to_waveform_end_to_end_vs_oracle():
    """End to end at the BASELINE sizes: the codes the HIP Dual-AR generates for the voice-clone case (== the
    reference's, asserted above) through the HIP codec, against the oracle codec on the fixture's codes: RMS <= 1e-4."""
    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.dual_ar import generate
    from oracle import dac as D

    z, skw = _load("s2_clone")
    cfg, model, _ = _model(skw)
    got = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]),
                   temperature=float(z["temperature"]), top_p=float(z["top_p"]), top_k=int(z["top_k"]),
                   seed=int(z["uniform_seed"]))
    T = z["prompt"].shape[1]
    codes = got[1:, T:-1].unsqueeze(0).contiguous()            # generate_long's slice (inference.py:708)
    want_codes = torch.from_numpy(z["tokens"])[1:, T:-1].unsqueeze(0).contiguous()
    assert torch.equal(codes, want_codes)
    dcfg = D.DacConfig()
    dstate = D.make_synthetic_state(dcfg, seed=3)
    codec = MiDAC.from_state_dict(DacConfig.from_any(dcfg), dstate, device=DEV)
    wav = codec.from_indices(codes.clone().to(DEV)).cpu()
    ref = D.DacOracle(dcfg, dstate).from_indices(want_codes.clone())
    assert wav.shape == ref.shape == (1, 1, (int(z["max_new"]) - 1) * 2048)
    e = float((wav - ref).pow(2).mean().sqrt())
    print("S2 end to end: waveform RMS error vs oracle", e, "signal RMS", float(ref.pow(2).mean().sqrt()))
    assert e <= 1e-4


def test_s2_serve_stream_two_staggered_requests_equal_the_fixtures_and_the_offline_codec():
    """serving.serve_stream at the BASELINE width with the full-size codec (VERDICT r03 weak #1e): s2_plain is live
    when s2_clone arrives (a clock the test advances puts it a few frames later), both stream their audio in chunks
    while sharing the frame loop.  Per request the streamed codes are the reference's fixture and the concatenated
    segments equal the offline `from_indices` of those codes bit for bit."""
    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.serving import StreamRequest, collect, serve_stream
    from oracle import dac as D

    zp, skw = _load("s2_plain")
    zc, _ = _load("s2_clone")
    cfg, model, _ = _model(skw)
    dcfg = D.DacConfig()
    codec = MiDAC.from_state_dict(DacConfig.from_any(dcfg), D.make_synthetic_state(dcfg, seed=3), device=DEV)
    model.set_ignore_eos(False)
    now = [0.0]

    def clock():          # every look at the clock is 2 ms later: the second request is due a few iterations in
        now[0] += 0.002
        return now[0]

    reqs = [StreamRequest(prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]), seed=int(z["uniform_seed"]),
                          rid=i, arrival=a, temperature=float(z["temperature"]), top_p=float(z["top_p"]), top_k=int(z["top_k"]))
            for i, (z, a) in enumerate(((zp, 0.0), (zc, 0.05)))]
    evs = list(serve_stream(model=model, codec=codec, requests=reqs, max_batch=8, step_frames=4, first_chunk_frames=4,
                            chunk_frames=16, chunk_growth=1.5, clock=clock, wait=lambda s: None, admit_early=False))
    first = {e.rid: e.t_emit for e in evs if e.kind == "segment" and e.t0 == 0}
    assert first[0] < first[1], "the second request was meant to join a running loop"
    got = collect(evs, codec.frame_length)
    for i, z in enumerate((zp, zc)):
        T = z["prompt"].shape[1]
        want = torch.from_numpy(z["tokens"])[1:, T:-1]                      # inference.py:708
        audio, codes = got[i]
        # the codec clamps the indices it is handed IN PLACE to its codebook sizes like the reference (rvq.py:354-359;
        # this synthetic pairing has 4096-code fast codebooks in front of 1024-entry residual quantizers), and the
        # events carry what was voiced
        clamped = want.clone()
        clamped[0].clamp_(max=dcfg.semantic_codebook_size - 1)
        clamped[1:].clamp_(max=dcfg.codebook_size - 1)
        assert torch.equal(codes, clamped), f"request {i}: streamed codes differ from the reference's fixture"
        off = codec.from_indices(want[None].clone().to(DEV))[0, 0].cpu()
        assert torch.equal(audio, off), f"request {i}: streamed audio differs from the offline decode"
    codec.stream_reset()


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "dualar_s2_long.npz")), reason="fixture not generated")
def test_s2_long_prompt_crosses_the_mfma_decode_attention_threshold():
    """dualar_s2_long.npz (round 4): the unmodified reference's generate() on the 4.56 B hash-weight model with a
    1010-token voice-clone-shaped prompt, 24 greedy frames (every decision >= 16 bf16 steps of margin): frames from
    position 1024 on run the decode attention on attn_decode_mfma_kernel + attn_decode_merge_kernel at the BASELINE
    shape (G = 4, D = 128), the 1010-row prefill goes through the 256-column GEMM tiles and 16 KV pages.  The whole
    (11, 1034) matrix equals the reference's -- with the MFMA pair, and with the VALU kernel for every row."""
    from fish_speech_amd.dual_ar import generate

    z, skw = _load("s2_long")
    cfg, model, _ = _model(skw, max_seq_len=2048, slots=2)
    want = z["tokens"]
    T = z["prompt"].shape[1]
    assert T < 1024 < want.shape[1]
    kw = dict(prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]), temperature=float(z["temperature"]),
              top_p=float(z["top_p"]), top_k=int(z["top_k"]), seed=int(z["uniform_seed"]))
    got = generate(model=model, **kw).numpy()
    bad = np.argwhere(got != want) if got.shape == want.shape else None
    assert got.shape == want.shape and len(bad) == 0, f"first mismatch at {None if bad is None else bad[0].tolist()}"
    model.set_attn_long_threshold(0)
    got2 = generate(model=model, **kw).numpy()
    model.set_attn_long_threshold(1024)
    assert np.array_equal(got2, want), "VALU decode attention for every row differs"
    print("s2_long full sequence equal;", str(z["note"]))

"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors and, where the reference
checkout exists (authoring container), against the unmodified reference modules."""
import numpy as np
import pytest
import torch

from oracle import dual_ar as O
from oracle.refload import reference_available
from tests.helpers import load_dualar_case


@pytest.mark.parametrize("case", ["tiny", "mid"])
def test_oracle_teacher_forced_vs_reference_golden(case):
    """Every frame of the reference's greedy run, replayed through the oracle: floating-point taps
    within 3 bf16 steps of the stored reference traces, every decision a near-argmax of the reference
    logits (bit-exact wherever the margin allows).  Holds on any CPU; on the CPU that produced the
    fixtures it is bit-exact (next test)."""
    from tests.helpers import check_teacher_forced, oracle_step_fn

    cfg, state, z = load_dualar_case(case)
    st = check_teacher_forced(oracle_step_fn(cfg, state, int(z["uniform_seed"])), cfg, z)
    assert st["frames"] == z["greedy"].shape[1] - z["prompt"].shape[1]
    assert st["exact"] >= 0.9 * st["decisions"]


@pytest.mark.parametrize("case", ["tiny", "mid"])
@pytest.mark.parametrize("mode,top_k", [("greedy", 1), ("sampled", 30)])
def test_oracle_free_running_vs_reference_golden(case, mode, top_k):
    """Free-running generation.  The reference's bf16 CPU path is not bit-reproducible across CPU
    models (mkldnn picks different fp32 summation orders), so away from the machine that wrote the
    fixtures the comparison is required only up to the first decision whose reference margin is
    below 2 bf16 steps; on the authoring machine the whole sequence matches."""
    cfg, state, z = load_dualar_case(case)
    orc = O.DualAROracle(cfg, state)
    y = O.generate(orc, torch.from_numpy(z["prompt"]), int(z["max_new"]), 0.7, 0.7, top_k,
                   uniform_fn=O.FmiUniform(int(z["uniform_seed"]), 0)).numpy()
    want = z[mode]
    if y.shape == want.shape and np.array_equal(y, want):
        return
    assert mode == "greedy", "sampled trajectories are only comparable on the fixture's CPU"
    T = z["prompt"].shape[1]
    k = O.robust_prefix(torch.from_numpy(z["greedy_margins_ulps"]), 2.0)
    assert np.array_equal(y[:, : T + k], want[:, : T + k])


@pytest.mark.parametrize("case", ["tiny_peaky", "tiny_peaky_eos", "mid_peaky", "tiny_sampled", "tiny_peaky_int8", "mid_long"])
def test_well_conditioned_fixtures_full_sequence_and_margins(case):
    """The well-conditioned fixtures (oracle.make_peaky_state; written by the unmodified reference's generate()):
    every decision of the reference run has >= 8 bf16 steps of margin (greedy) -- re-derived here from the stored
    reference traces --, so ANY correct bf16 implementation must reproduce the WHOLE token sequence, on any CPU:
    the oracle does, free-running (greedy and sampled), and teacher-forced with exact == decisions."""
    from tests.helpers import bf16_from_u16, check_teacher_forced, oracle_step_fn

    cfg, state, z = load_dualar_case(case)
    if case.endswith("_int8"):     # the reference ran its WeightOnlyInt8QuantHandler over this state
        state = O.quantize_state_int8(cfg, state)
    want = z["tokens"]
    T = z["prompt"].shape[1]
    n = want.shape[1] - T
    top_k = int(z["top_k"])
    assert n >= 40 or case == "mid_long"
    if top_k == 1:
        m = O.greedy_frame_margins(cfg, bf16_from_u16(z["slow_logits_live"]), bf16_from_u16(z["fast_logits"]))
        assert float(m.min()) >= 8.0 and O.robust_prefix(m, 8.0) == n
        assert np.array_equal(m.numpy(), z["greedy_margins_ulps"])
    if case == "mid_long":   # a 1010-token prompt: the margins above are the CPU check; the oracle's free run equalled the
        return               # reference's when the fixture was written (oracle/gen_golden.py asserts it) -- minutes of CPU here
    y = O.generate(O.DualAROracle(cfg, state), torch.from_numpy(z["prompt"]), int(z["max_new"]), float(z["temperature"]),
                   float(z["top_p"]), top_k, uniform_fn=O.FmiUniform(int(z["uniform_seed"]), 0)).numpy()
    assert y.shape == want.shape and np.array_equal(y, want)
    if case == "tiny_peaky_eos":
        assert n < int(z["max_new"]) and int(want[0, -1]) == cfg.im_end_id
    if top_k == 1:
        st = check_teacher_forced(oracle_step_fn(cfg, state, int(z["uniform_seed"])), cfg, z)
        assert st["frames"] == n and st["exact"] == st["decisions"] == n * cfg.num_codebooks


def test_generate_bounds_match_reference_errors():
    cfg = O.DualARConfig()
    orc = O.DualAROracle(cfg, O.make_synthetic_state(cfg, 0))
    with pytest.raises(ValueError):  # inference.py:263-266
        O.generate(orc, O.make_prompt(cfg, cfg.max_seq_len, 1), 4)


def test_sampler_top_k1_is_seed_invariant():
    torch.manual_seed(0)
    lg = (torch.randn(1, 1, 300) * 3).bfloat16()
    t, p = torch.tensor(0.7).bfloat16(), torch.tensor(0.7).bfloat16()
    want = int(lg[0, 0].float().argmax())
    for seed in range(5):
        u = O.FmiUniform(seed)
        got = int(O.sample(lg, t, p, 1, u))
        # u == 0 makes the reference's race return index 0 (documented quirk, 1/256 per draw)
        assert got in (want, 0)


def test_uniform_generator_is_byte_uniformish():
    b = O.fmi_uniform_u8(1, 2, 3, 4, 1 << 16)
    assert b.dtype == np.uint8 and 120 < b.mean() < 135 and len(np.unique(b)) == 256


@pytest.mark.skipif(not reference_available(), reason="reference checkout not present on this box")
def test_oracle_bit_exact_vs_reference_fp32_and_bf16():
    from oracle.gen_golden import _ref_generate

    cfg = O.DualARConfig()
    for dtype in (torch.float32, torch.bfloat16):
        st = O.make_synthetic_state(cfg, seed=3, dtype=dtype, head_gain=8.0)
        prompt = O.make_prompt(cfg, 17, seed=4, n_semantic=5)
        ref = _ref_generate(cfg, st, prompt, 10, 30, O.FmiUniform(99))
        got = O.generate(O.DualAROracle(cfg, st), prompt, 10, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(99))
        assert torch.equal(ref, got.long())


PROJIN = dict(fast_dim=96, fast_n_head=3, fast_n_local_heads=1, fast_head_dim=32, fast_intermediate_size=192)


def test_oracle_fast_project_in_bit_exact_vs_reference():
    """fast_dim != dim: the reference puts a Linear(dim, fast_dim) WITH bias between the two transformers
    (llama.py:665-668,827); the oracle's restatement equals the unmodified reference's generate() bit for bit."""
    from oracle.gen_golden import _ref_generate

    cfg = O.DualARConfig(**PROJIN)
    assert cfg.has_fast_project_in
    for dtype in (torch.float32, torch.bfloat16):
        st = O.make_synthetic_state(cfg, seed=5, dtype=dtype, head_gain=8.0)
        assert st["fast_project_in.weight"].shape == (96, 128) and st["fast_project_in.bias"].shape == (96,)
        prompt = O.make_prompt(cfg, 19, seed=6, n_semantic=6)
        ref = _ref_generate(cfg, st, prompt, 10, 30, O.FmiUniform(77))
        got = O.generate(O.DualAROracle(cfg, st), prompt, 10, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(77))
        assert torch.equal(ref, got.long())


def test_int8_weight_only_oracle_vs_reference_golden():
    """§8f #2 groundwork: the reference's weight-only int8 path (tools/llama/quantize.py:186-229 -- per-row
    symmetric quantisation, `F.linear(x, w.to(bf16)) * scales`) restated in the oracle, against the fixture the
    unmodified reference wrote after quantising the tiny model with its own handler
    (oracle/gen_golden.py asserts the restated quantiser reproduces that checkpoint bit for bit)."""
    from tests.helpers import check_teacher_forced, oracle_step_fn

    cfg, state, z = load_dualar_case("tiny_int8")
    q = O.quantize_state_int8(cfg, state)
    n_int8 = [k for k, v in q.items() if v.dtype == torch.int8]
    assert len(n_int8) == 5 * (cfg.n_layer + cfg.n_fast_layer) + 1 and "embeddings.weight" not in n_int8
    for k in n_int8:   # per-row grid with s = max|row| / 127.5: |w - q*s| <= s/2, up to s for the clipped row maximum
        # (127.5 rounds to 128, clamps to 127), plus 127 * 2^-9 * s from the stored scale being bf16
        w, s = state[k].float(), q[k[:-6] + "scales"].float()
        assert int(q[k].abs().max()) <= 128 and bool(((w - q[k].float() * s[:, None]).abs() <= 1.3 * s[:, None] + 1e-6).all())
    st = check_teacher_forced(oracle_step_fn(cfg, q, int(z["uniform_seed"])), cfg, z)
    assert st["frames"] == z["greedy"].shape[1] - z["prompt"].shape[1]
    assert st["exact"] >= 0.9 * st["decisions"]
    _, _, zb = load_dualar_case("tiny")       # quantisation is visible: the traces differ from the bf16 model's
    assert not np.array_equal(z["slow_logits_live"], zb["slow_logits_live"])


def test_int4_group_quantiser_restatement_equals_the_reference_functions():
    """Groundwork for the int4 weight-only format (quantize.py:52-163,300-349; refused by the product path): the
    restated group quantiser gives, bit for bit, the reference's 4-bit values, packed scales_and_zeros and
    dequantised weights, including the handler's zero-padding of in_features to a multiple of 1024.  (The CUDA-only
    tile shuffle `_convert_weight_to_int4pack` and the mm arithmetic cannot be run here and stay unpinned.)"""
    from oracle.refload import add_reference_to_path

    add_reference_to_path()
    from tools.llama import quantize as RQ

    torch.manual_seed(3)
    for shape, gs in (((96, 256), 128), ((64, 1024), 32), ((40, 512), 256)):
        w = (torch.randn(shape) * 0.05).bfloat16()
        q, sz = O.quantize_int4_groups(w, gs)
        rq, rsz = RQ.group_quantize_tensor(w, n_bit=4, groupsize=gs)
        assert torch.equal(q, rq) and torch.equal(sz, rsz) and sz.shape == (shape[1] // gs, shape[0], 2)
        assert int(q.min()) >= 0 and int(q.max()) <= 15
        assert torch.equal(O.dequantize_int4_groups(q, sz.float(), gs), RQ.group_dequantize_tensor(rq, rsz.float(), 4, gs))
        err = (O.dequantize_int4_groups(q, sz.float(), gs) - w.float()).abs().reshape(-1, gs).amax(1)
        assert bool((err <= sz[..., 0].float().t().reshape(-1) * 0.51 + 1e-3).all())       # half a step of the group's grid
    # the handler's padding rule on a whole (tiny) checkpoint: in_features 128 / 256 are not multiples of 1024 -> no
    # padding needed only when groupsize and 128 divide them
    cfg = O.DualARConfig()
    st = O.make_synthetic_state(cfg, seed=1)
    q4 = O.quantize_state_int4(cfg, st)
    assert "layers.0.attention.wqkv.weight" not in q4 and q4["layers.0.attention.wqkv.weight_int4"].dtype == torch.int32
    assert q4["layers.0.attention.wqkv.weight_int4"].shape[1] == cfg.dim            # 128: divisible by 128, no padding
    assert q4["embeddings.weight"].dtype == torch.bfloat16 and "embeddings.scales_and_zeros" not in q4
    assert RQ._check_linear_int4_k(cfg.dim, 128, 8) and not RQ._check_linear_int4_k(96, 128, 8)
    w96 = (torch.randn(16, 96) * 0.05).bfloat16()
    assert O.quantize_state_int4(cfg, {"x.weight": w96})["x.weight_int4"].shape == (16, 1024)   # padded like the handler


MUTATION_CASES = ["tiny_peaky", "tiny_peaky_eos", "mid_peaky", "tiny_sampled", "tiny_projin", "tiny", "mid"]


@pytest.mark.parametrize("case", MUTATION_CASES)
def test_parity_fixtures_detect_injected_numeric_faults(case):
    """VERDICT r04 #4a -- the parity suite must be able to FAIL for the right reasons.  Five arithmetic faults are injected
    into the oracle (every FFN down-projection x 0.5, ONE layer's down-projection x 0.5, RoPE not applied, fast attention
    contributing nothing, slow attention contributing nothing); every fixture family must notice each of them through
    at least one of the checks the GPU suite runs on it: the free-running token matrix, or the teacher-forced float taps
    at check_teacher_forced's tolerances.  The unmodified oracle trips neither.  (Token equality ALONE misses most of
    these on the well-conditioned fixtures -- by construction their decisions ride on the embedding -> tied-head path
    --, which is why every family, the sampled one included since round 5, also carries the reference's float traces;
    the table is committed as profiles/r05_mutation_table.txt, tools/mutation_table.py.)"""
    from tests.mutations import MUTATIONS, detect

    clean = detect(case, None)
    assert clean["tokens_changed"] == 0 and clean["taps_fail"] is False, clean
    for m in MUTATIONS:
        r = detect(case, m)
        assert r["tokens_changed"] != 0 or r["taps_fail"], f"{case} does not notice '{m}': {r}"
        assert r["taps_fail"], f"{case}: the float taps miss '{m}' ({r})"

"""GPU parity tests of the Dual-AR path: every call goes through the C ABI (libfishmi.so) and is
checked against the CPU oracle (oracle/dual_ar.py, itself pinned to the reference's goldens).

Tolerances (stated per north_star): codebook / token indices bit-exact; floating-point taps (logits,
hidden) within 2 bf16 ulps (+1e-3 absolute) of the oracle -- the model computes in bf16 with fp32
accumulation and only the fp32 summation ORDER differs from the CPU path."""
import ctypes as C

import os

import numpy as np
import pytest
import torch

from oracle import dual_ar as O
from tests.helpers import bf16_close, load_dualar_case

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    from fish_speech_amd import _lib

    return _lib.load()


def _linear(lib, x, w, norm_w, res, M, N, K, epi, path, eps=1e-6):
    from fish_speech_amd._lib import check

    n_out = N // 2 if epi == 2 else N
    out = torch.zeros(M, n_out, dtype=torch.bfloat16, device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    nd = norm_w.to(DEV) if norm_w is not None else None
    rd = res.to(DEV) if res is not None else None
    check(lib.fmi_op_linear_bf16(C.c_void_p(xd.data_ptr()), C.c_void_p(wd.data_ptr()),
                                 C.c_void_p(nd.data_ptr()) if nd is not None else None,
                                 C.c_void_p(rd.data_ptr()) if rd is not None else None,
                                 C.c_void_p(out.data_ptr()), M, N, K, eps, epi, path, None))
    torch.cuda.synchronize()
    return out.cpu()


def _linear_oracle(x, w, norm_w, res, epi, eps=1e-6):
    import torch.nn.functional as F

    xin = O.rms_norm(x, norm_w, eps) if norm_w is not None else x
    if epi == 2:
        h = w.shape[0] // 2
        return F.silu(F.linear(xin, w[:h])) * F.linear(xin, w[h:])
    y = F.linear(xin, w)
    return res + y if epi == 1 else y


@pytest.mark.parametrize("M,path", [(1, 1), (3, 1), (8, 1), (16, 1), (5, 2), (17, 2), (130, 2), (300, 2)])
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("norm", [False, True])
def test_linear_kernels_match_oracle(lib, M, path, epi, norm):
    g = torch.Generator().manual_seed(M * 7 + epi * 3 + int(norm))
    N, K = (192, 256) if epi != 2 else (2 * 160, 128)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16() if norm else None
    res = torch.randn(M, N, generator=g).bfloat16() if epi == 1 else None
    got = _linear(lib, x, w, nw, res, M, N, K, epi, path)
    want = _linear_oracle(x, w, nw, res, epi)
    ok, mx, nbad = bf16_close(got, want, scale=res)  # res + y may cancel: tolerance follows the operands
    assert ok, f"max abs err {mx}, {nbad} elements out of tolerance"


@pytest.mark.parametrize("N,K", [(6144, 2560), (2560, 9728), (4112, 2560)])
def test_skinny_linear_s2_shapes_and_batch_invariance(lib, N, K):
    """Asymmetric data at the S2-Pro GEMV shapes; row b of a batch-8 call must equal the batch-1 call
    bit for bit (the batch lives in the MFMA N dimension)."""
    g = torch.Generator().manual_seed(N)
    x = torch.randn(8, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    full = _linear(lib, x, w, None, None, 8, N, K, 0, 1)
    ok, mx, nbad = bf16_close(full, _linear_oracle(x, w, None, None, 0))
    assert ok, (mx, nbad)
    one = _linear(lib, x[3:4].contiguous(), w, None, None, 1, N, K, 0, 1)
    assert torch.equal(one[0], full[3])


@pytest.mark.parametrize("N,K,epi,norm", [(19456, 2560, 2, True), (6144, 2560, 0, True), (2560, 4096, 1, False),
                                          (2560, 9728, 1, False), (4112, 2560, 0, True), (2560, 2080, 1, True),
                                          (192, 256, 0, True), (320, 96, 2, False)])
def test_skinny_linear_row_result_is_independent_of_the_batch_size(lib, N, K, epi, norm):
    """Up to 8 rows the GEMV fetches the activation fragments of a k-tile pair with one all-lanes load and keeps
    the two tiles' sums in the two column halves of the MFMA; above 8 rows it loads per tile.  Both must give every
    row the same bits (same products, same order): a row computed alone, in a batch of 8 and in a batch of 13,
    at the S2-Pro shapes, an odd tile count, an odd number of k-tiles (2080 = 65, 96 = 3) and tiny shapes."""
    g = torch.Generator().manual_seed(N * 3 + K)
    x = torch.randn(13, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16() if norm else None
    res = torch.randn(13, N if epi != 2 else N // 2, generator=g).bfloat16() if epi == 1 else None
    big = _linear(lib, x, w, nw, res, 13, N, K, epi, 1)
    ok, mx, nbad = bf16_close(big, _linear_oracle(x, w, nw, res, epi), scale=res)
    assert ok, (mx, nbad)
    eight = _linear(lib, x[:8].contiguous(), w, nw, None if res is None else res[:8].contiguous(), 8, N, K, epi, 1)
    assert torch.equal(eight, big[:8])
    for r in (0, 5, 12):
        one = _linear(lib, x[r:r + 1].contiguous(), w, nw, None if res is None else res[r:r + 1].contiguous(), 1, N, K, epi, 1)
        assert torch.equal(one[0], big[r]), r


@pytest.mark.parametrize("N,K,epi,norm", [(2560, 4096, 1, False), (2560, 9728, 1, False), (6144, 2560, 0, True)])
@pytest.mark.parametrize("M", [1, 5, 8])
def test_row_balanced_decode_gemv_is_bit_identical_to_the_16_row_tiles(lib, N, K, epi, norm, M):
    """The row-balanced decode copy (10 / 12 weight rows per tile so that 256 equal work-groups cover N = 2560 /
    6144: dualar_kernels.h skinny_row_plan) holds the same products in the same order as the 16-row tiling: every
    output equals the 16-row kernel's bit for bit, for any batch <= 8; shapes without a balanced variant are refused
    by the forced path (the model then streams the 16-row tiles)."""
    from fish_speech_amd import FishmiError

    g = torch.Generator().manual_seed(N + K + M)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16() if norm else None
    res = torch.randn(M, N, generator=g).bfloat16() if epi == 1 else None
    tiles16 = _linear(lib, x, w, nw, res, M, N, K, epi, 1)
    rows = _linear(lib, x, w, nw, res, M, N, K, epi, 6)
    assert torch.equal(rows, tiles16)
    ok, mx, nbad = bf16_close(rows, _linear_oracle(x, w, nw, res, epi), scale=res)
    assert ok, (mx, nbad)
    if M == 8 and epi == 0:
        with pytest.raises(FishmiError):   # 4096 rows = 16 per CU: the 16-row tiling is the balanced one
            _linear(lib, x, w[:4096].contiguous(), nw, None, M, 4096, K, 0, 6)


@pytest.mark.parametrize("M,N,K,epi", [(1600, 6144, 2560, 0), (1600, 2560, 4096, 1), (777, 19456, 2560, 2),
                                        (130, 2560, 9728, 1), (5, 192, 256, 0), (300, 320, 96, 2), (129, 4112, 2080, 0),
                                        (2500, 2560, 4096, 1), (260, 2576, 2560, 0)])
def test_lds_staged_prefill_gemm_equals_the_direct_variant(lib, M, N, K, epi):
    """The prefill GEMM stages its operands through LDS (DMA, double-buffered); the earlier variant feeds the
    MFMAs straight from L2.  Same MFMA order per output element -> identical bits, at the S2-Pro shapes (8 x 200
    prompt rows), ragged M / N tails and an odd number of k-tiles; and both match the oracle."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    res = torch.randn(M, N, generator=g).bfloat16() if epi == 1 else None
    staged = _linear(lib, x, w, None, res, M, N, K, epi, 7)     # LDS-staged, 4 waves
    direct = _linear(lib, x, w, None, res, M, N, K, epi, 5)
    assert torch.equal(staged, direct), float((staged.float() - direct.float()).abs().max())
    # wave-specialised variant (4 compute + 4 loader waves; falls back to the 4-wave kernel for an odd k-tile count)
    ws = _linear(lib, x, w, None, res, M, N, K, epi, 8)
    assert torch.equal(ws, direct), float((ws.float() - direct.float()).abs().max())
    # its 128 x 256-tile form (8 compute + 4 loader waves, three stages), taken by shape once it fills the chip
    wide = _linear(lib, x, w, None, res, M, N, K, epi, 9)
    assert torch.equal(wide, direct), float((wide.float() - direct.float()).abs().max())
    # round 4: 256-column tiles, one 8-wave work-group per CU, products deferred across the barrier (256 / 128 / 64 / 192 rows),
    # and the 16-wave / plain forms kept for A/B; all stage whole 128-byte activation lines and store through LDS
    for path in (10, 11, 12, 13, 14, 15):
        y = _linear(lib, x, w, None, res, M, N, K, epi, path)
        assert torch.equal(y, direct), (path, float((y.float() - direct.float()).abs().max()))
    picked = _linear(lib, x, w, None, res, M, N, K, epi, 2)   # whichever the shape selects
    if 1024 <= N <= 2560 and K >= 4096:
        # round 6: the wo / w2 shapes run with the contraction split in three (linear_tiled_ksplit: by (N, K) only, so that a
        # row's bits never depend on the rows it travels with) -- another fp32 summation order than the variants above:
        # within the oracle's tolerance, and the first rows of this call equal a call of those rows alone, bit for bit
        ok, mx, nbad = bf16_close(picked, _linear_oracle(x, w, None, res, epi), scale=res)
        assert ok, (mx, nbad)
        few = _linear(lib, x[:17].contiguous(), w, None, res[:17].contiguous() if res is not None else None, 17, N, K, epi, 2)
        assert torch.equal(few, picked[:17]), "split contraction: a row's bits depend on the row count of the call"
    else:
        assert torch.equal(picked, direct)
    if M <= 300:
        ok, mx, nbad = bf16_close(staged, _linear_oracle(x, w, None, res, epi), scale=res)
        assert ok, (mx, nbad)


def _sample(lib, logits, ids, samp, frame, draw, prev, sem):
    from fish_speech_amd._lib import SamplingC, check

    B, n = logits.shape
    ld = n
    out = torch.zeros(B, dtype=torch.int32, device=DEV)
    lg = logits.to(DEV).contiguous()
    idd = ids.to(DEV).int().contiguous() if ids is not None else None
    pv = prev.to(DEV).int().contiguous() if prev is not None else None
    sp = SamplingC(*samp)
    check(lib.fmi_op_sample(C.c_void_p(lg.data_ptr()), B, n, ld, C.c_void_p(idd.data_ptr()) if idd is not None else None,
                            C.byref(sp), frame, draw, C.c_void_p(pv.data_ptr()) if pv is not None else None,
                            sem[0], sem[1], C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    return out.cpu()


def _oracle_draw(row_logits, ids, vocab, temperature, top_p, top_k, seed, stream, frame, draw):
    """inference.py:54-93 on a full-vocab row with -inf outside the live ids."""
    full = torch.full((vocab,), float("-inf"), dtype=torch.bfloat16)
    full[ids.long()] = row_logits
    u = O.FmiUniform(seed, stream)
    u.frame, u.draw_idx = frame, draw
    t = torch.tensor(temperature).bfloat16()
    p = torch.tensor(top_p).bfloat16()
    return int(O.sample(full[None, None], t, p, top_k, u))


@pytest.mark.parametrize("top_k", [1, 30, 200])
@pytest.mark.parametrize("n", [64, 257, 4097])
def test_sampler_bit_exact_vs_oracle(lib, top_k, n):
    g = torch.Generator().manual_seed(n + top_k)
    B, vocab = 8, n + 100
    ids = torch.sort(torch.randperm(vocab, generator=g)[:n]).values.int()
    for gain in (1.0, 4.0):
        logits = (torch.randn(B, n, generator=g) * gain).bfloat16()  # many exact bf16 ties
        got = _sample(lib, logits, ids, (0.7, 0.7, top_k, 77, 0), 5, 3, None, (0, 0))
        for b in range(B):
            want = _oracle_draw(logits[b], ids, vocab, 0.7, 0.7, top_k, 77, 0, 5, 3)   # stream keyed by seed only
            assert int(got[b]) == want, (b, int(got[b]), want)


@pytest.mark.parametrize("top_k", [1, 30, 64])
def test_sampler_candidate_selection_paths_bit_exact_on_adversarial_rows(lib, top_k):
    """Round 5: the top-k candidates are found by counting keys per distance below the row maximum (32 buckets of 16 bf16
    key units) -- one shared list when at most 64 keys lie inside the reach that holds k, per-wave lists when more do, the
    radix descent per wave when a wave alone holds more than 64 or the reach (four octaves) does not hold k.  Rows built
    to hit each path and each boundary, against the oracle's draw: a lone peak over noise far below (descent), all
    logits equal (4097-way tie: descent, ties by index), exactly k / k + 1 / 64 / 65 / 300 keys inside one bucket of the
    maximum, every logit negative, a maximum at the top of the bf16 range of interest, -inf entries, and n not a
    multiple of the work-group size."""
    g = torch.Generator().manual_seed(1000 + top_k)
    n, vocab = 4097, 4300
    ids = torch.sort(torch.randperm(vocab, generator=g)[:n]).values.int()
    rows = []
    base = torch.randn(n, generator=g) * 0.5 - 20.0                      # noise far below everything that follows
    r = base.clone(); r[123] = 30.0; rows.append(r)                      # lone peak: the reach never holds k > 1
    rows.append(torch.full((n,), 1.5))                                   # every key equal
    for m in (top_k, top_k + 1, 64, 65, 300):                            # m keys within a few steps of the maximum
        r = base.clone()
        idx = torch.randperm(n, generator=g)[:m]
        r[idx] = 8.0 - 0.03125 * torch.randint(0, 12, (m,), generator=g).float()
        rows.append(r)
    rows.append(-(torch.rand(n, generator=g) * 3 + 0.5))                 # all negative, dense near the maximum
    r = torch.randn(n, generator=g) * 40.0; rows.append(r)               # wide: the k-th largest is octaves below the max
    r = torch.randn(n, generator=g); r[::3] = float("-inf"); rows.append(r)
    logits = torch.stack(rows).bfloat16()
    for nn in (n, 4000, 2049):
        lg = logits[:, :nn].contiguous()
        idn = ids[:nn].contiguous()
        got = _sample(lib, lg, idn, (0.7, 0.9, top_k, 4321, 0), 9, 1, None, (0, 0))
        for b in range(lg.shape[0]):
            want = _oracle_draw(lg[b], idn, vocab, 0.7, 0.9, top_k, 4321, 0, 9, 1)
            assert int(got[b]) == want, (nn, b, int(got[b]), want)


def test_sampler_log_table_matches_torch_bf16():
    """-log(u) for the 256 possible uniforms, in bf16, equals torch's CPU result (inference.py:44-45);
    checked through single-candidate draws: u == 0 is the only case returning token 0."""
    u = torch.arange(256, dtype=torch.float32) / 256
    q = -torch.log(u.bfloat16())
    assert torch.isinf(q[0]) and bool((q[1:] > 0).all())


def test_sampler_ras_selection(lib):
    """Repetition-aware sampling (inference.py:118-144): the high-temperature draw replaces the normal
    one iff the normal token is semantic and sits in the 10-frame window."""
    g = torch.Generator().manual_seed(3)
    B, n = 16, 300
    ids = torch.arange(100, 100 + n).int()
    logits = (torch.randn(B, n, generator=g) * 3).bfloat16()
    base = _sample(lib, logits, ids, (0.7, 0.7, 30, 5, 0), 2, 0, None, (0, 0))
    prev = torch.zeros(B, 10, dtype=torch.int32)
    prev[:, 4] = base  # the normal draw is "in the window" for every row
    sem = (120, 350)
    got = _sample(lib, logits, ids, (0.7, 0.7, 30, 5, 1), 2, 0, prev, sem)
    for b in range(B):
        tn = _oracle_draw(logits[b], ids, 500, 0.7, 0.7, 30, 5, 0, 2, 0)
        th = _oracle_draw(logits[b], ids, 500, 1.0, 0.9, 30, 5, 0, 2, 1)
        want = th if (sem[0] <= tn <= sem[1]) else tn
        assert int(base[b]) == tn and int(got[b]) == want


# ------------------------------------------------------------------------------- whole frame step


def _make_model(cfg, state, max_batch=4):
    from fish_speech_amd.dual_ar import MiDualAR

    m = MiDualAR.from_state_dict(cfg, state, device=DEV, im_end_id=cfg.im_end_id)
    m.setup_caches(max_batch, cfg.max_seq_len)
    return m


def hip_step_fn(model, cfg, uniform_seed, temperature=0.7, top_p=0.7, top_k=1):
    """The HIP path (fmi_dualar_step through the C ABI) behind the check_teacher_forced protocol."""
    model.set_trace(True)

    def step(f, x, pos0, prev):
        sp = model._sampling(temperature, top_p, top_k, uniform_seed, prev is not None)
        out = model.step(x.to(DEV), pos0, sp, prev.to(DEV) if prev is not None else None, f).cpu()
        logits, ids, hidden, _ = model.debug_taps(1)
        tr = model.fast_trace(1)[0].cpu()
        return out, logits[0].cpu(), hidden[0].cpu(), tr[1:]

    return step


@pytest.mark.parametrize("case", ["tiny", "mid"])
def test_teacher_forced_frames_match_reference_golden(case):
    """The decode_one_token seam fed with the REFERENCE's token history (tests/golden): every
    floating-point tap of every frame within 3 bf16 steps of the reference's trace, every decision a
    near-argmax of the reference's logits -- bit-exact indices wherever the reference's margin exceeds
    the rounding noise of a different fp32 summation order."""
    from tests.helpers import check_teacher_forced

    cfg, state, z = load_dualar_case(case)
    model = _make_model(cfg, state)
    st = check_teacher_forced(hip_step_fn(model, cfg, int(z["uniform_seed"])), cfg, z)
    print(case, st)
    assert st["frames"] == z["greedy"].shape[1] - z["prompt"].shape[1]
    assert st["exact"] >= 0.9 * st["decisions"], st


@pytest.mark.parametrize("case", ["tiny", "mid"])
def test_generate_free_running_vs_reference_golden(case):
    """Free-running greedy generation (prefill + hipGraph decode) against the reference's output: must
    be identical at least up to the first decision whose reference margin is below 2 bf16 steps (past
    that point any two correct implementations may diverge; the reference does so across CPUs)."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, z = load_dualar_case(case)
    model = _make_model(cfg, state)
    got = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]),
                   temperature=0.7, top_p=0.7, top_k=1, seed=int(z["uniform_seed"])).numpy()
    want = z["greedy"]
    T = z["prompt"].shape[1]
    k = O.robust_prefix(torch.from_numpy(z["greedy_margins_ulps"]), 2.0)
    assert got.shape == want.shape
    agree = int(np.argmin((got == want).all(axis=0))) if not (got == want).all() else got.shape[1]
    print(case, "robust prefix frames", k, "agreement columns", agree - T, "of", want.shape[1] - T)
    assert np.array_equal(got[:, : T + k], want[:, : T + k])
    assert np.array_equal(got[:, :T], want[:, :T])


@pytest.mark.parametrize("case", ["tiny_peaky", "tiny_peaky_eos", "mid_peaky", "tiny_sampled", "tiny_projin", "mid_long"])
def test_generate_full_sequence_equals_reference_on_well_conditioned_fixtures(case):
    """STRICT index parity: prefill + hipGraph decode loop, free-running, against the token
    sequence the UNMODIFIED reference's generate() (inference.py:243-359) wrote for the well-conditioned fixtures
    (48-64 frames; every reference decision has >= 8 bf16 steps of margin, the sampled run is invariant under 2 steps
    of logit noise at every decision): the WHOLE sequence must be equal -- greedy, greedy ending by <|im_end|>, the
    10-codebook mid model, and sampled (top-k 30, top-p, temperature, RAS firing 17 times)."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, z = load_dualar_case(case)
    model = _make_model(cfg, state)
    want = z["tokens"]
    got = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]),
                   temperature=float(z["temperature"]), top_p=float(z["top_p"]), top_k=int(z["top_k"]),
                   seed=int(z["uniform_seed"])).numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"first mismatch at (row, column) {bad[0].tolist()}: got {got[tuple(bad[0])]}, want {want[tuple(bad[0])]}"
    # and through the eager (no hipGraph) path, with EOS polled every frame
    model.set_graph(False)
    got2 = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]),
                    temperature=float(z["temperature"]), top_p=float(z["top_p"]), top_k=int(z["top_k"]),
                    seed=int(z["uniform_seed"]), poll_every=1).numpy()
    assert np.array_equal(got2, want)
    if case == "mid_long":
        # round 4: a 1010-token prompt + 40 frames -- frames 14.. sit at positions >= 1024, where decode attention runs on
        # attn_decode_mfma_kernel + attn_decode_merge_kernel; the reference's indices must come out of BOTH regimes
        # (threshold 0 = the fused VALU kernel for every row)
        assert z["prompt"].shape[1] < 1024 < want.shape[1]
        model.set_graph(True)
        model.set_attn_long_threshold(0)
        got3 = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]),
                        temperature=float(z["temperature"]), top_p=float(z["top_p"]), top_k=int(z["top_k"]),
                        seed=int(z["uniform_seed"])).numpy()
        model.set_attn_long_threshold(1024)
        assert np.array_equal(got3, want)


@pytest.mark.parametrize("case", ["tiny_peaky", "tiny_peaky_eos", "mid_peaky", "tiny_projin", "mid_long"])
def test_teacher_forced_exact_decisions_on_well_conditioned_fixtures(case):
    """Teacher-forced through the decode_one_token seam with the reference's history: taps within the bf16 noise
    bound AND every single decision equal to the reference's (exact == decisions)."""
    from tests.helpers import check_teacher_forced

    cfg, state, z = load_dualar_case(case)
    model = _make_model(cfg, state)
    st = check_teacher_forced(hip_step_fn(model, cfg, int(z["uniform_seed"])), cfg, z)
    print(case, st)
    n = z["tokens"].shape[1] - z["prompt"].shape[1]
    assert st["frames"] == n and st["exact"] == st["decisions"] == n * cfg.num_codebooks, st


def test_teacher_forced_taps_and_equal_draws_on_the_sampled_fixture():
    """tiny_sampled (top-k 30, top-p 0.9, temperature 0.7; 5 draws leave the top-1 candidate, RAS fires 17 times): its
    token equality alone is a weak detector of numeric faults (tests/mutations.py: 2 of 5 injected faults move a token),
    so the reference's FLOAT traces are checked too -- teacher-forced with the fixture's own sampling parameters, every
    tap within the bf16 noise bound and every draw equal to the reference's."""
    from tests.helpers import check_teacher_forced

    cfg, state, z = load_dualar_case("tiny_sampled")
    model = _make_model(cfg, state)
    st = check_teacher_forced(hip_step_fn(model, cfg, int(z["uniform_seed"]), float(z["temperature"]), float(z["top_p"]),
                                          int(z["top_k"])), cfg, z, decide="equal")
    print("tiny_sampled", st)
    n = z["tokens"].shape[1] - z["prompt"].shape[1]
    assert st["frames"] == n and st["exact"] == st["decisions"] == n * cfg.num_codebooks, st


def test_stream_priority_change_recreates_the_stream_and_leaves_generation_unchanged():
    """fmi_dualar_set_stream_priority (round 5): the handle's private stream is re-created with another dispatch priority,
    the captured frame graphs are dropped -- a scheduling knob, the tokens do not move."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, z = load_dualar_case("tiny_peaky")
    model = _make_model(cfg, state)
    kw = dict(prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]), temperature=float(z["temperature"]),
              top_p=float(z["top_p"]), top_k=int(z["top_k"]), seed=int(z["uniform_seed"]))
    want = z["tokens"]
    for prio in (-1, 1, 0):
        model.set_stream_priority(prio)
        assert np.array_equal(generate(model=model, **kw).numpy(), want), prio


def test_generate_matches_oracle_run_on_this_box():
    """Same comparison against the oracle run on the GPU box's own CPU (tiny case, greedy)."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, z = load_dualar_case("tiny")
    orc = O.DualAROracle(cfg, state)
    orc.trace = {}
    want = O.generate(orc, torch.from_numpy(z["prompt"]), 12, 0.7, 0.7, 1, uniform_fn=O.FmiUniform(1234, 0)).numpy()
    model = _make_model(cfg, state)
    got = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=12, temperature=0.7, top_p=0.7,
                   top_k=1, seed=1234).numpy()
    ids = torch.from_numpy(z["live_ids"]).long()
    slow = torch.stack(orc.trace["slow_logits"])[:, ids]
    fast = torch.stack([torch.stack(f) for f in orc.trace["fast_logits"]])
    k = O.robust_prefix(O.greedy_frame_margins(cfg, slow, fast), 2.0)
    T = z["prompt"].shape[1]
    assert np.array_equal(got[:, : T + k], want[:, : T + k])


def test_batch_equals_single_utterance_and_graph_equals_eager():
    """Batch > 1 is new capability: each utterance of a ragged batch must equal its batch-1 run
    bit for bit, and hipGraph replay must equal eager launches."""
    from fish_speech_amd.dual_ar import generate, generate_batch

    cfg, state, z = load_dualar_case("mid")
    prompts = [O.make_prompt(cfg, T, seed=s, n_semantic=ns) for T, s, ns in ((40, 9, 12), (23, 10, 0), (57, 11, 20))]
    model = _make_model(cfg, state, max_batch=4)
    seeds = [11, 12, 13]
    batch = generate_batch(model=model, prompts=prompts, max_new_tokens=12, temperature=0.7, top_p=0.7, top_k=30,
                           seeds=seeds)
    model.set_graph(False)
    for i, p in enumerate(prompts):
        # the uniform stream is keyed by the seed only: run each utterance alone with its seed
        single = generate_batch(model=model, prompts=[prompts[0]] * i + [p], max_new_tokens=12, temperature=0.7,
                                top_p=0.7, top_k=30, seeds=seeds[: i + 1])[-1]
        assert torch.equal(single, batch[i]), i


def test_im_end_stops_generation_and_bounds():
    from fish_speech_amd.dual_ar import generate

    cfg, state, _ = load_dualar_case("tiny")
    # make <|im_end|> the certain winner: its embedding row aligned with everything
    st = dict(state)
    e = st["embeddings.weight"].clone()
    e[cfg.im_end_id] = e[cfg.semantic_begin_id: cfg.semantic_end_id + 1].float().mean(0).bfloat16() * 0
    st["embeddings.weight"] = e
    model = _make_model(cfg, st)
    with pytest.raises(ValueError):  # inference.py:263-266
        generate(model=model, prompt=O.make_prompt(cfg, cfg.max_seq_len, 1), max_new_tokens=4)
    out = generate(model=model, prompt=O.make_prompt(cfg, 10, 2), max_new_tokens=cfg.max_seq_len, top_k=1,
                   temperature=0.7, top_p=0.7, poll_every=4)
    assert out.shape[1] <= cfg.max_seq_len  # clamp of inference.py:268-275


# ------------------------------------------------------------------------------- BASELINE-size properties


@pytest.fixture(scope="module")
def s2_model():
    """S2-Pro-shaped 4.56 B parameter model with random weights generated on the GPU (bench.py's)."""
    import bench
    from fish_speech_amd.dual_ar import MiDualAR

    cfg = bench.s2_pro_config(max_seq_len=512)
    model = MiDualAR(cfg, device=DEV, im_end_id=cfg.im_end_id)
    model.load_state_dict(bench.synthetic_state_on_device(cfg, torch.device(DEV)))
    model.setup_caches(8, 512)
    model.set_ignore_eos(True)
    return cfg, model


def test_s2_shape_ragged_batch_equals_single_and_graph_equals_eager(s2_model):
    """Size-independent properties at the BASELINE model size (36+4 layers, dim 2560, vocab 155 776):
    a ragged batch of 8 (config 4: mixed-length prompts) gives, for every utterance, exactly the
    tokens of its batch-1 run; hipGraph replay equals eager launches; all codes are in range."""
    from fish_speech_amd.dual_ar import generate_batch

    cfg, model = s2_model
    lens = [200, 57, 400, 131, 64, 333, 200, 90]
    prompts = []
    for i, T in enumerate(lens):
        g = torch.Generator().manual_seed(100 + i)
        p = torch.zeros(cfg.num_codebooks + 1, T, dtype=torch.int64)
        p[0] = torch.randint(0, 150000, (T,), generator=g)
        if i % 2:  # voice-clone shaped: semantic tail with codes
            ns = T // 3
            codes = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks, ns), generator=g)
            p[1:, T - ns:] = codes
            p[0, T - ns:] = codes[0] + cfg.semantic_begin_id
        prompts.append(p)
    seeds = list(range(50, 58))
    kw = dict(max_new_tokens=10, temperature=0.7, top_p=0.7, top_k=30, stop_on_im_end=False)
    model.set_graph(True)
    batch = generate_batch(model=model, prompts=prompts, seeds=seeds, **kw)
    model.set_graph(False)
    eager = generate_batch(model=model, prompts=prompts, seeds=seeds, **kw)
    model.set_graph(True)
    for i in range(8):
        assert batch[i].shape == (cfg.num_codebooks + 1, lens[i] + 10)
        assert torch.equal(batch[i], eager[i]), f"graph != eager for utterance {i}"
        gen = batch[i][:, lens[i]:]
        assert int(gen[1:].min()) >= 0 and int(gen[1:].max()) < cfg.codebook_size
        tok = gen[0]
        ok = ((tok >= cfg.semantic_begin_id) & (tok <= cfg.semantic_end_id)) | (tok == cfg.im_end_id) | (tok == 0)
        assert bool(ok.all()), "slow token outside the constrained set (semantic ids, <|im_end|>, or the u==0 token 0)"
    for i in (2, 5):  # batch-1 runs (the slot does not matter: the uniform stream is keyed by the seed)
        single = generate_batch(model=model, prompts=[prompts[0]] * i + [prompts[i]], seeds=seeds[: i + 1], **kw)[-1]
        assert torch.equal(single, batch[i]), f"batch result of utterance {i} differs from its batch-1 run"


def test_s2_shape_device_codes_feed_codec_shape(s2_model):
    from fish_speech_amd.dual_ar import generate_batch_device

    cfg, model = s2_model
    prompts = [torch.zeros(cfg.num_codebooks + 1, 20, dtype=torch.int64) for _ in range(3)]
    for i, p in enumerate(prompts):
        p[0] = torch.randint(0, 150000, (20,), generator=torch.Generator().manual_seed(i))
    codes = generate_batch_device(model=model, prompts=prompts, max_new_tokens=7, seeds=[1, 2, 3], temperature=0.7,
                                  top_p=0.7, top_k=30)
    assert codes.shape == (3, cfg.num_codebooks, 7) and codes.dtype == torch.int64 and codes.is_cuda
    assert int(codes.min()) >= 0 and int(codes.max()) < cfg.codebook_size


# ------------------------------------------------------------------------------- model-object seam


class _SeamAdapter:
    """Lets the oracle's restatement of decode_one_token_ar (inference.py:96-181) drive the HIP model
    through the model-object seam: forward_generate / forward_generate_fast / fast_embeddings."""

    def __init__(self, model, cfg):
        self.m, self.cfg, self.trace = model, cfg, None

    def forward_generate(self, x, input_pos, math_backend):
        r = self.m.forward_generate(x.to(DEV), input_pos.to(DEV))
        return r.logits.cpu(), r.hidden_states.cpu()

    def forward_generate_fast(self, h, pos):
        return self.m.forward_generate_fast(h.to(DEV), pos).cpu()

    def fast_embeddings(self, a):
        return self.m.fast_embeddings(a.to(DEV)).cpu()


@pytest.mark.parametrize("case,top_k", [("tiny", 30), ("mid", 30), ("mid", 1)])
def test_reference_decode_function_over_hip_model_equals_hip_step(case, top_k):
    """The reference's decode_one_token_ar logic (oracle restatement: torch sort/softmax/cumsum sampler,
    RAS, fast chain) running on the HIP model's forward_generate/forward_generate_fast must produce,
    frame after frame, exactly the tokens of the fused HIP step (fmi_dualar_step): same logits, and a
    HIP sampler that is bit-exact to the reference's sampling semantics."""
    from fish_speech_amd.dual_ar import decode_one_token

    cfg, state, z = load_dualar_case(case)
    model = _make_model(cfg, state)
    ad = _SeamAdapter(model, cfg)
    prompt = torch.from_numpy(z["prompt"])
    T = prompt.shape[1]
    ncb1 = cfg.num_codebooks + 1
    bias = O.semantic_logit_bias(cfg, torch.bfloat16)
    temp = torch.tensor(0.7).bfloat16()
    u = O.FmiUniform(4321, 0)
    window = torch.zeros(ncb1, 10, dtype=torch.int32)
    cur, pos = prompt.view(1, ncb1, -1), torch.arange(T)
    n_same = 0
    for f in range(12):
        u.frame, u.draw_idx = f, 0
        prev = None if f == 0 else window.clone()
        want = O.decode_one_token(ad, cur, pos, temp, temp, top_k, bias, prev, u, math_backend=True).view(-1)
        sp = model._sampling(0.7, 0.7, top_k, 4321, prev is not None)
        xs = cur.reshape(ncb1, -1).t().int().contiguous().to(DEV)
        got = model.step(xs, int(pos[0]), sp, prev.to(DEV) if prev is not None else None, f).cpu()
        assert torch.equal(got.long(), want.long()), f"frame {f}: fused step {got.tolist()} != seam path {want.tolist()}"
        n_same += 1
        if f > 0:
            window = window.roll(-1, dims=1)
            window[:, -1] = want.int()
        cur, pos = want.view(1, ncb1, 1).long(), torch.tensor([T + f])
    assert n_same == 12


# ------------------------------------------------------------------------------- edge cases / misuse


@pytest.mark.parametrize("T", [1, 63, 64, 65, 130])
def test_prompt_lengths_around_kv_page_boundaries_vs_oracle(T):
    """KV pages hold 64 tokens: prompts of 1, 63, 64, 65 and 130 tokens (0, 1 and 2 page crossings, then
    decode frames that cross the next boundary) against the oracle run on this box: logits within the
    bf16 noise bound at every frame, tokens equal wherever the oracle's margin allows."""
    cfg, state, z = load_dualar_case("tiny")
    model = _make_model(cfg, state)
    prompt = O.make_prompt(cfg, T, seed=T, n_semantic=min(T // 2, 20))
    orc = O.DualAROracle(cfg, state)
    orc.trace = {}
    n_new = 5
    seq = O.generate(orc, prompt, n_new, 0.7, 0.7, 1, uniform_fn=O.FmiUniform(99, 0), stop_on_im_end=False)
    ids = torch.from_numpy(z["live_ids"]).long()
    ncb1 = cfg.num_codebooks + 1
    window = torch.zeros(ncb1, 10, dtype=torch.int32)
    for f in range(n_new):
        if f == 0:
            x, pos0, prev = prompt.t().int().contiguous(), 0, None
        else:
            x, pos0, prev = seq[:, T + f - 1].view(1, ncb1).int().contiguous(), T + f - 1, window.clone()
        sp = model._sampling(0.7, 0.7, 1, 99, prev is not None)
        model.step(x.to(DEV), pos0, sp, prev.to(DEV) if prev is not None else None, f)
        logits, _, hidden, _ = model.debug_taps(1)
        want = orc.trace["slow_logits"][f][ids].float()
        rel = float((logits[0].float().cpu() - want).norm() / want.norm())
        assert rel <= 2e-2, f"T={T} frame {f}: relative L2 error {rel:.4f}"
        if f > 0:
            window = window.roll(-1, dims=1)
            window[:, -1] = seq[:, T + f].int()


def test_single_frame_and_full_length_generation():
    from fish_speech_amd.dual_ar import generate

    cfg, state, _ = load_dualar_case("tiny")
    model = _make_model(cfg, state)
    model.set_ignore_eos(True)
    p = O.make_prompt(cfg, 9, seed=3)
    one = generate(model=model, prompt=p, max_new_tokens=1, top_k=1, temperature=0.7, top_p=0.7, seed=5)
    assert one.shape == (cfg.num_codebooks + 1, 10)  # only the prefill frame (inference.py:336-347 with n-1 = 0)
    # max_new_tokens = 0 means "fill up to max_seq_len" (inference.py:276-278)
    full = generate(model=model, prompt=p, max_new_tokens=0, top_k=1, temperature=0.7, top_p=0.7, seed=5,
                    stop_on_im_end=False)
    assert full.shape[1] == cfg.max_seq_len
    assert torch.equal(full[:, :10], one)
    # a prompt of max_seq_len - 1 tokens can still produce exactly one frame
    p2 = O.make_prompt(cfg, cfg.max_seq_len - 1, seed=4)
    out = generate(model=model, prompt=p2, max_new_tokens=50, top_k=1, temperature=0.7, top_p=0.7, seed=5)
    assert out.shape[1] == cfg.max_seq_len


def test_c_abi_misuse_raises_instead_of_crashing():
    from fish_speech_amd import FishmiError
    from fish_speech_amd.dual_ar import MiDualAR, generate_batch

    cfg, state, _ = load_dualar_case("tiny")
    bare = MiDualAR(cfg, device=DEV, im_end_id=cfg.im_end_id)
    bare.setup_caches(1, cfg.max_seq_len)
    with pytest.raises(FishmiError):  # weights never loaded
        bare.prefill([0], [O.make_prompt(cfg, 4, 1)], [4], [bare._sampling(0.7, 0.7, 1, 0)])
    model = _make_model(cfg, state, max_batch=2)
    with pytest.raises(FishmiError):  # slot outside the cache
        model.prefill([5], [O.make_prompt(cfg, 4, 1)], [4], [model._sampling(0.7, 0.7, 1, 0)])
    with pytest.raises(ValueError):   # more utterances than slots
        generate_batch(model=model, prompts=[O.make_prompt(cfg, 4, i) for i in range(3)], max_new_tokens=2)
    with pytest.raises(FishmiError):  # wrong tensor shape at load time
        bad = dict(state)
        bad["layers.0.attention.wo.weight"] = bad["layers.0.attention.wo.weight"][:-16]
        MiDualAR.from_state_dict(cfg, bad, device=DEV, im_end_id=cfg.im_end_id)
    with pytest.raises(FishmiError):  # a tensor missing at finalize
        miss = {k: v for k, v in state.items() if k != "fast_norm.weight"}
        MiDualAR.from_state_dict(cfg, miss, device=DEV, im_end_id=cfg.im_end_id)
    with pytest.raises(FishmiError):  # no CPU fallback
        MiDualAR(cfg, device="cpu", im_end_id=cfg.im_end_id)


def test_s2_shape_forward_passes_match_the_cpu_oracle(s2_model):
    """Full-width check (4.56 B parameters: the GEMV variants, D=128 G=4 attention, tiled prefill GEMM and paged
    KV that the small fixtures do not reach): slow forward over a 12-token prompt, one decode-position slow
    forward on top of its KV cache, and two fast-transformer steps, teacher-forced against the CPU oracle.

    After 36 layers two valid fp32 summation orders of a bf16 model are ~3 % apart (they drift like
    sqrt(layers): 1.3 % after the fixtures' 3 layers), so the tolerance is calibrated instead of guessed: the
    oracle is also run with fp32 weights and activations (no bf16 rounding = the exact model), and the HIP
    path must be as close to that exact result as the bf16 CPU restatement is (<= 1.3x its relative L2 and
    <= 1.5x its worst element), and no further from the bf16 oracle than twice that noise."""
    import dataclasses

    import bench

    cfg, model = s2_model
    state = {k: v.cpu() for k, v in bench.synthetic_state_on_device(cfg, torch.device(DEV)).items()}
    ocfg = O.DualARConfig(**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(O.DualARConfig)
                             if hasattr(cfg, f.name)})
    ocfg.max_seq_len = 64
    orc = O.DualAROracle(ocfg, state)
    orc.setup_caches(1, 64)
    exact = O.DualAROracle(ocfg, {k: v.float() for k, v in state.items()})
    exact.setup_caches(1, 64)
    ids = model._table(1, torch.int32).view(-1).long().cpu()          # live LM-head rows

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    def close(got, want, ideal, what):
        got, want, ideal = (t.float().cpu().reshape(-1) for t in (got, want, ideal))
        noise, noise_max = rel(want, ideal), float((want - ideal).abs().max())
        e_ideal, e_max, e_orc = rel(got, ideal), float((got - ideal).abs().max()), rel(got, want)
        print(f"S2 shape, {what}: vs exact fp32 model: HIP {e_ideal:.4f} (max {e_max:.4f}), "
              f"bf16 CPU oracle {noise:.4f} (max {noise_max:.4f}); HIP vs oracle {e_orc:.4f}")
        assert e_ideal <= 1.3 * noise + 1e-3, what
        assert e_max <= 1.5 * noise_max + 1e-3, what
        assert e_orc <= 2.0 * noise + 1e-3, what

    T = 12
    g = torch.Generator().manual_seed(77)
    x = torch.zeros(1, cfg.num_codebooks + 1, T, dtype=torch.int64)
    x[0, 0] = torch.randint(0, 150000, (T,), generator=g)
    codes = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks, 4), generator=g)
    x[0, 1:, T - 4:] = codes                                            # voice-clone shaped tail
    x[0, 0, T - 4:] = codes[0] + cfg.semantic_begin_id
    want_logits, want_hidden = orc.forward_generate(x, torch.arange(T), math_backend=True)
    ex_logits, ex_hidden = exact.forward_generate(x, torch.arange(T), math_backend=True)
    r = model.forward_generate(x.to(DEV), torch.arange(T, device=DEV))
    close(r.logits[0, 0].cpu()[ids], want_logits[0, 0][ids], ex_logits[0, 0][ids], "prefill logits")
    close(r.hidden_states, want_hidden, ex_hidden, "prefill hidden")

    frame = torch.zeros(1, cfg.num_codebooks + 1, 1, dtype=torch.int64)
    frame[0, 1:, 0] = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks,), generator=g)
    frame[0, 0, 0] = frame[0, 1, 0] + cfg.semantic_begin_id
    want_logits, want_hidden = orc.forward_generate(frame, torch.tensor([T]), math_backend=True)
    ex_logits, ex_hidden = exact.forward_generate(frame, torch.tensor([T]), math_backend=True)
    r = model.forward_generate(frame.to(DEV), torch.tensor([T], device=DEV))
    close(r.logits[0, 0].cpu()[ids], want_logits[0, 0][ids], ex_logits[0, 0][ids], "decode-position logits")
    close(r.hidden_states, want_hidden, ex_hidden, "decode-position hidden")

    h = want_hidden.reshape(1, -1)                                       # teacher-forced fast chain (4 layers)
    for pos in range(2):
        want_fast = orc.forward_generate_fast(h, torch.tensor([pos]))
        ex_fast = exact.forward_generate_fast(h.float(), torch.tensor([pos]))
        got_fast = model.forward_generate_fast(h.to(DEV), torch.tensor([pos], device=DEV))
        close(got_fast, want_fast, ex_fast, f"fast logits at codebook position {pos}")
        a = torch.tensor([int(want_fast.reshape(-1).float().argmax())])
        want_e = orc.fast_embeddings(a)
        assert torch.equal(model.fast_embeddings(a).cpu(), want_e)
        h = want_e.reshape(1, -1)


def test_s2_shape_random_weights_prompt_of_250_and_decode_positions_across_a_page_boundary(s2_model):
    """VERDICT r04 #4b: the float check at the BASELINE width beyond one KV page, on RANDOM weights (no peaky structure:
    every layer's arithmetic decides these taps) and with the calibrated criterion instead of a guessed bound.  A
    250-token voice-clone-shaped prompt (four KV pages, tiled prefill GEMM + MFMA flash attention over 250 positions),
    then ten teacher-forced decode positions 250..259 -- the decode GEMVs and the paged decode attention, crossing the
    page boundary at 256 -- each compared with the bf16 CPU oracle AND the fp32-exact oracle: the HIP path must be as
    close to the exact model as the reference's own bf16 CPU arithmetic is (<= 1.3 x its relative L2 per tap, <= 1.5 x
    its worst element over all decode taps together)."""
    import dataclasses

    import bench

    cfg, model = s2_model
    dev_state = bench.synthetic_state_on_device(cfg, torch.device(DEV))
    w2_dev = {k: v for k, v in dev_state.items() if k.startswith("layers.") and k.endswith("feed_forward.w2.weight")}
    state = {k: v.cpu() for k, v in dev_state.items()}
    del dev_state
    ocfg = O.DualARConfig(**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(O.DualARConfig)
                             if hasattr(cfg, f.name)})
    ocfg.max_seq_len = 320
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(nthreads, 32))
    try:
        orc = O.DualAROracle(ocfg, state)
        orc.setup_caches(1, 320)
        exact = O.DualAROracle(ocfg, {k: v.float() for k, v in state.items()})
        exact.setup_caches(1, 320)
        del state
        ids = model._table(1, torch.int32).view(-1).long().cpu()

        def rel(a, b):
            return float((a - b).norm() / b.norm())

        worst = {"hip": 0.0, "orc": 0.0}

        def close(got, want, ideal, what, decode):
            got, want, ideal = (t.float().cpu().reshape(-1) for t in (got, want, ideal))
            noise, e_ideal, e_orc = rel(want, ideal), rel(got, ideal), rel(got, want)
            if decode:
                worst["hip"] = max(worst["hip"], float((got - ideal).abs().max() / ideal.abs().max()))
                worst["orc"] = max(worst["orc"], float((want - ideal).abs().max() / ideal.abs().max()))
            print(f"S2 random weights, {what}: vs exact: HIP {e_ideal:.4f}, bf16 oracle {noise:.4f}; HIP vs oracle {e_orc:.4f}")
            assert e_ideal <= 1.3 * noise + 1e-3, what
            assert e_orc <= 2.0 * noise + 1e-3, what

        T, ND = 250, 10
        g = torch.Generator().manual_seed(2025)
        x = torch.zeros(1, cfg.num_codebooks + 1, T, dtype=torch.int64)
        x[0, 0] = torch.randint(0, 150000, (T,), generator=g)
        codes = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks, 100), generator=g)
        x[0, 1:, T - 100:] = codes
        x[0, 0, T - 100:] = codes[0] + cfg.semantic_begin_id
        want_logits, want_hidden = orc.forward_generate(x, torch.arange(T), math_backend=True)
        ex_logits, ex_hidden = exact.forward_generate(x, torch.arange(T), math_backend=True)
        r = model.forward_generate(x.to(DEV), torch.arange(T, device=DEV))
        close(r.logits[0, 0].cpu()[ids], want_logits[0, 0][ids], ex_logits[0, 0][ids], "prefill(250) logits", False)
        close(r.hidden_states, want_hidden, ex_hidden, "prefill(250) hidden", False)
        for pos in range(T, T + ND):
            frame = torch.zeros(1, cfg.num_codebooks + 1, 1, dtype=torch.int64)
            frame[0, 1:, 0] = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks,), generator=g)
            frame[0, 0, 0] = frame[0, 1, 0] + cfg.semantic_begin_id
            want_logits, want_hidden = orc.forward_generate(frame, torch.tensor([pos]), math_backend=True)
            ex_logits, ex_hidden = exact.forward_generate(frame, torch.tensor([pos]), math_backend=True)
            r = model.forward_generate(frame.to(DEV), torch.tensor([pos], device=DEV))
            close(r.logits[0, 0].cpu()[ids], want_logits[0, 0][ids], ex_logits[0, 0][ids], f"position {pos} logits", True)
            close(r.hidden_states, want_hidden, ex_hidden, f"position {pos} hidden", True)
        print("worst element over the decode taps (relative to the tap's largest): HIP", worst["hip"], "oracle", worst["orc"])
        assert worst["hip"] <= 1.5 * worst["orc"] + 1e-3
        # The criterion can FAIL (tests/mutations.py, on the HIP side): the judge's own mutation -- every slow FFN
        # down-projection x 0.5 -- loaded into the HIP model must be refused at the next position; the original
        # weights restored, the same position passes again (and the shared model is left as it was found).
        pos = T + ND
        frame = torch.zeros(1, cfg.num_codebooks + 1, 1, dtype=torch.int64)
        frame[0, 1:, 0] = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks,), generator=g)
        frame[0, 0, 0] = frame[0, 1, 0] + cfg.semantic_begin_id
        want_logits, want_hidden = orc.forward_generate(frame, torch.tensor([pos]), math_backend=True)
        ex_logits, ex_hidden = exact.forward_generate(frame, torch.tensor([pos]), math_backend=True)
        try:   # (the shared model gets its weights back even if a check below fails: ADVICE r05)
            model.load_state_dict({k: (v.float() * 0.5).to(v.dtype) for k, v in w2_dev.items()})
            r = model.forward_generate(frame.to(DEV), torch.tensor([pos], device=DEV))
            with pytest.raises(AssertionError):
                close(r.hidden_states, want_hidden, ex_hidden, f"position {pos} hidden, every slow w2 halved", False)
            with pytest.raises(AssertionError):
                close(r.logits[0, 0].cpu()[ids], want_logits[0, 0][ids], ex_logits[0, 0][ids], f"position {pos} logits, every slow w2 halved", False)
        finally:
            model.load_state_dict(w2_dev)
        r = model.forward_generate(frame.to(DEV), torch.tensor([pos], device=DEV))
        close(r.logits[0, 0].cpu()[ids], want_logits[0, 0][ids], ex_logits[0, 0][ids], f"position {pos} logits, weights restored", False)
        close(r.hidden_states, want_hidden, ex_hidden, f"position {pos} hidden, weights restored", False)
    finally:
        torch.set_num_threads(nthreads)


def test_s2_shape_fast_chain_every_codebook_position_on_the_frame_loop_path(s2_model):
    """the float check of the FAST chain at the BASELINE width at ALL of its positions, on the path the
    benchmark runs.  Eight (hidden, frame) pairs -- normed hidden rows taken from the model's own slow forward over eight
    short prompts, so they are real hidden states -- go through `fast_chain_forced`: tail() as a decode frame runs it
    at batch 8 (merged positions 0/1, batch GEMV, the tabulated layer-0 q|k|v), every draw replaced by the frame the
    bf16 CPU oracle's argmax dictates (teacher forcing: a near-tie cannot fork the chain).  The same rows run through
    the bf16 oracle and the fp32-exact oracle.  At each of positions 1..9 (position 0's logits do not exist: the
    reference discards them, inference.py:148-149; its K/V feed every later position) the HIP logits must be as close
    to the exact model as the reference's bf16 CPU arithmetic is (<= 1.3 x its relative L2, rows pooled), with the
    table ON and OFF and merged / two-pass; ON == OFF and merged == two-pass bit for bit.  A fault injected into the
    fast layers of the HIP model (every fast w2 x 0.5) must be refused."""
    import dataclasses

    import bench

    cfg, model = s2_model
    B, ncb = 8, cfg.num_codebooks
    dev_state = bench.synthetic_state_on_device(cfg, torch.device(DEV))
    fast_keys = [k for k in dev_state if k.startswith("fast_")]
    w2_dev = {k: dev_state[k] for k in fast_keys if k.endswith("feed_forward.w2.weight")}
    state = {k: dev_state[k].cpu() for k in fast_keys}
    state["embeddings.weight"] = torch.zeros(1, cfg.dim, dtype=torch.bfloat16)     # (dtype marker only; no slow pass here)
    del dev_state
    ocfg = O.DualARConfig(**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(O.DualARConfig)
                             if hasattr(cfg, f.name)})
    ocfg.n_layer, ocfg.max_seq_len = 0, 16
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(nthreads, 32))
    try:
        orc = O.DualAROracle(ocfg, state)
        orc.setup_caches(B, 16)
        exact = O.DualAROracle(ocfg, {k: v.float() for k, v in state.items()})
        exact.setup_caches(B, 16)
        # real hidden states: the slow forward of eight 6-token prompts (its parity is the two tests above)
        g = torch.Generator().manual_seed(606)
        hidden = []
        for i in range(B):
            x = torch.zeros(1, ncb + 1, 6, dtype=torch.int64)
            x[0, 0] = torch.randint(0, 150000, (6,), generator=g)
            hidden.append(model.forward_generate(x.to(DEV), torch.arange(6, device=DEV)).hidden_states.reshape(-1).cpu())
        hidden = torch.stack(hidden)                                                  # (B, dim) bf16
        code0 = torch.randint(0, cfg.codebook_size, (B,), generator=g)
        # oracles, teacher-forced by the bf16 oracle's argmax
        orc.forward_generate_fast(hidden, torch.tensor([0]))
        exact.forward_generate_fast(hidden.float(), torch.tensor([0]))
        forced = torch.zeros(B, ncb + 1, dtype=torch.int32)
        forced[:, 0] = (code0 + cfg.semantic_begin_id).int()
        forced[:, 1] = code0.int()
        want, ideal, a = [], [], code0
        for cb in range(1, ncb):
            e = orc.fast_embeddings(a)
            want.append(orc.forward_generate_fast(e, torch.tensor([cb])).reshape(B, -1))
            ideal.append(exact.forward_generate_fast(e.float(), torch.tensor([cb])).reshape(B, -1))
            a = want[-1].float().argmax(dim=-1)
            forced[:, 1 + cb] = a.int()
        del orc, exact, state

        slots = list(range(B))
        prompts = [torch.zeros(ncb + 1, 4, dtype=torch.int64) for _ in range(B)]

        def run(table, merge):
            model.set_fast_merge(merge)
            model.prefill(slots, prompts, [4] * B, [model._sampling(0.7, 0.7, 1, 0)] * B)
            out = model.fast_chain_forced(hidden, forced, slots, table=table).cpu()
            for sl in slots:
                model.release(sl)
            return out

        def rel(x, y):
            return float((x.float() - y.float()).norm() / y.float().norm())

        def check(got, what):
            for cb in range(1, ncb):
                w, i, gt = want[cb - 1], ideal[cb - 1], got[:, cb]
                noise, e_ideal, e_orc = rel(w, i), rel(gt, i), rel(gt, w)
                print(f"S2 fast chain ({what}), position {cb}: vs exact: HIP {e_ideal:.4f}, bf16 oracle {noise:.4f}; HIP vs oracle {e_orc:.4f}")
                assert e_ideal <= 1.3 * noise + 1e-3, (what, cb)
                assert e_orc <= 2.0 * noise + 1e-3, (what, cb)

        got_on = run(True, True)
        assert model.derived_info()["table_rows"] == cfg.codebook_size, "the table is not in use: this would not test the benchmark's path"
        check(got_on, "table on, merged")
        got_off = run(False, True)
        check(got_off, "table off, merged")
        assert torch.equal(got_on[:, 1:], got_off[:, 1:]), "tabulated layer-0 q|k|v changes bits"
        got_two = run(True, False)
        assert torch.equal(got_on[:, 1:], got_two[:, 1:]), "merged positions 0/1 change bits"
        # fault injection on the HIP side, restored whatever happens
        try:
            model.load_state_dict({k: (v.float() * 0.5).to(v.dtype) for k, v in w2_dev.items()})
            bad = run(True, True)
            with pytest.raises(AssertionError):
                check(bad, "every fast w2 halved")
        finally:
            model.load_state_dict(w2_dev)
            model.set_fast_merge(True)
        check(run(True, True), "weights restored")
    finally:
        torch.set_num_threads(nthreads)


MISS_ULPS_S2 = 8.0


def test_s2_shape_teacher_forced_index_agreement_vs_oracle_trace(s2_model):
    """Index parity at the BASELINE model size, ASSERTED: the CPU oracle free-runs 16 greedy frames of the S2-Pro
    shaped random-weight model on this box (its trace = reference-equivalent logits, tests/test_oracle_cpu.py), the
    HIP path is driven with that token history through the decode_one_token seam, and every one of its 160 decisions
    (16 frames x (1 slow + 9 fast)) is compared with the oracle's:
      * agreement rate >= 80 % (random N(0, 0.02) weights: ~Gaussian logits, so about one decision in eight is a
        near-tie of the oracle itself -- see make_peaky_state for why no seed avoids them; measured on MI355X:
        76 of 87 compared decisions equal, the 11 misses at oracle margins 0,0,0,1,1,1,2,2,3,3,5 bf16 steps);
      * every miss lands on a token whose ORACLE logit is within 8 bf16 steps of the oracle's maximum -- the margin
        below which the well-conditioned fixtures call a decision unclear (after 36 bf16 layers two fp32 summation
        orders differ by a few steps per logit, test_s2_shape_forward_passes_match_the_cpu_oracle) -- never an
        outlier.  (A frame's fast chain is compared up to its first miss: a different code legitimately changes
        the rest of that frame.)"""
    import dataclasses

    import bench
    from tests.helpers import check_teacher_forced

    cfg, model = s2_model
    state = {k: v.cpu() for k, v in bench.synthetic_state_on_device(cfg, torch.device(DEV)).items()}
    ocfg = O.DualARConfig(**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(O.DualARConfig)
                             if hasattr(cfg, f.name)})
    ocfg.max_seq_len = 64
    T, n_frames = 12, 16
    g = torch.Generator().manual_seed(123)
    prompt = torch.zeros(cfg.num_codebooks + 1, T, dtype=torch.int64)
    prompt[0] = torch.randint(0, 150000, (T,), generator=g)
    codes = torch.randint(0, cfg.codebook_size, (cfg.num_codebooks, 5), generator=g)
    prompt[1:, T - 5:] = codes
    prompt[0, T - 5:] = codes[0] + cfg.semantic_begin_id
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(nthreads, 32))
    try:
        orc = O.DualAROracle(ocfg, state)
        orc.trace = {}
        seq = O.generate(orc, prompt, n_frames, 0.7, 0.7, 1, uniform_fn=O.FmiUniform(4242, 0), stop_on_im_end=False)
    finally:
        torch.set_num_threads(nthreads)
    ids = model._table(1, torch.int32).view(-1).long().cpu()
    u16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)

    class Z(dict):
        files = ["tokens"]

    z = Z(tokens=seq.numpy(), prompt=prompt.numpy(), live_ids=ids.numpy(),
          slow_logits_live=u16(torch.stack(orc.trace["slow_logits"])[:, ids]), hidden=u16(torch.stack(orc.trace["hidden"])),
          fast_logits=u16(torch.stack([torch.stack(f) for f in orc.trace["fast_logits"]])))
    st = check_teacher_forced(hip_step_fn(model, ocfg, 4242), ocfg, z, ulps=1e9, decide_ulps=1e9, rel_l2=0.08)
    model.set_trace(False)
    print("S2 shape teacher-forced:", st)
    assert st["frames"] == n_frames
    assert st["decisions"] >= 60 and st["exact"] >= 0.8 * st["decisions"], st
    assert max(st["miss_margins"], default=0.0) <= MISS_ULPS_S2, st


# ------------------------------------------------------------------------------- weight-only int8 (8f #2)


def _linear_int8(lib, x, wq, sc, norm_w, res, M, N, K, epi, stream_int8, eps=1e-6):
    from fish_speech_amd._lib import check

    n_out = N // 2 if epi == 2 else N
    out = torch.zeros(M, n_out, dtype=torch.bfloat16, device=DEV)
    t = [v.to(DEV) if v is not None else None for v in (x, wq, sc, norm_w, res)]
    ptr = lambda v: C.c_void_p(v.data_ptr()) if v is not None else None
    check(lib.fmi_op_linear_int8(ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), ptr(t[4]), C.c_void_p(out.data_ptr()),
                                 M, N, K, eps, epi, stream_int8, None))
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("N,K,epi,norm", [(19456, 2560, 2, True), (6144, 2560, 0, True), (2560, 4096, 1, False),
                                          (2560, 9728, 1, False), (4096, 2560, 0, True), (192, 128, 0, True),
                                          (320, 256, 2, False)])
@pytest.mark.parametrize("M", [1, 8, 13])
def test_int8_gemv_equals_the_dequantised_bf16_gemv_and_the_reference_arithmetic(lib, N, K, epi, norm, M):
    """WeightOnlyInt8Linear (tools/llama/quantize.py:204-229): bf16(bf16(x @ W_int8^T) * scales).  Streaming the
    int8 tiles and converting in registers gives the same bits as streaming the dequantised bf16 tiles (int8 ->
    bf16 is exact), for 1 / 8 rows (int8 stream) and 13 rows (bf16 stream of the same model); and both match the
    reference arithmetic evaluated on the CPU."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(N + K + M + epi)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    wq, sc = O.quantize_int8_per_channel(w)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16() if norm else None
    res = torch.randn(M, N, generator=g).bfloat16() if epi == 1 else None
    a = _linear_int8(lib, x, wq, sc, nw, res, M, N, K, epi, 1)
    b = _linear_int8(lib, x, wq, sc, nw, res, M, N, K, epi, 0)
    assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    xin = O.rms_norm(x, nw, 1e-6) if norm else x
    lin = lambda rows, srows: F.linear(xin, rows.to(torch.bfloat16)) * srows
    if epi == 2:
        h = N // 2
        want = F.silu(lin(wq[:h], sc[:h])) * lin(wq[h:], sc[h:])
    else:
        y = lin(wq, sc)
        want = res + y if epi == 1 else y
    ok, mx, nbad = bf16_close(a, want, scale=res)
    assert ok, (mx, nbad)
    if M > 1:
        one = _linear_int8(lib, x[:1].contiguous(), wq, sc, nw, None if res is None else res[:1].contiguous(), 1, N, K, epi, 1)
        assert torch.equal(one[0], a[0])


def test_int8_generate_full_sequence_equals_the_reference_int8_run():
    """Strict index parity of the weight-only-int8 path: the well-conditioned tiny model quantised by the REFERENCE's
    WeightOnlyInt8QuantHandler, its generate() run (48 frames, every decision >= 8 bf16 steps of margin) against the HIP
    path loaded from the same int8 checkpoint -- the whole sequence must be equal, and every teacher-forced decision."""
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate
    from tests.helpers import check_teacher_forced

    cfg, state, z = load_dualar_case("tiny_peaky_int8")
    q = O.quantize_state_int8(cfg, state)
    mcfg = DualARConfig.from_any(cfg)
    mcfg.weight_int8 = True
    model = MiDualAR(mcfg, device=DEV, im_end_id=cfg.im_end_id).load_state_dict(q)
    model.setup_caches(2, cfg.max_seq_len)
    got = generate(model=model, prompt=torch.from_numpy(z["prompt"]), max_new_tokens=int(z["max_new"]), temperature=0.7,
                   top_p=0.7, top_k=1, seed=int(z["uniform_seed"])).numpy()
    assert got.shape == z["tokens"].shape and np.array_equal(got, z["tokens"])
    st = check_teacher_forced(hip_step_fn(model, cfg, int(z["uniform_seed"])), cfg, z)
    n = z["tokens"].shape[1] - z["prompt"].shape[1]
    assert st["frames"] == n and st["exact"] == st["decisions"] == n * cfg.num_codebooks, st


def test_int8_model_vs_the_reference_int8_fixture():
    """The tiny model quantised by the reference's WeightOnlyInt8QuantHandler (fixture dualar_tiny_int8.npz): the
    HIP path loaded from the int8 checkpoint (int8 tiles for decode, dequantised tiles for prefill, scales in the
    epilogues), teacher-forced against the reference's traces, and free-running against the int8 oracle."""
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate
    from tests.helpers import check_teacher_forced

    cfg, state, z = load_dualar_case("tiny_int8")
    q = O.quantize_state_int8(cfg, state)
    mcfg = DualARConfig.from_any(cfg)
    mcfg.weight_int8 = True
    model = MiDualAR(mcfg, device=DEV, im_end_id=cfg.im_end_id).load_state_dict(q)
    model.setup_caches(2, cfg.max_seq_len)
    st = check_teacher_forced(hip_step_fn(model, cfg, int(z["uniform_seed"])), cfg, z)
    assert st["frames"] == z["greedy"].shape[1] - z["prompt"].shape[1] and st["exact"] >= 0.9 * st["decisions"]
    with pytest.raises(ValueError):      # an int8 checkpoint needs the int8 arena
        MiDualAR(DualARConfig.from_any(cfg), device=DEV, im_end_id=cfg.im_end_id).load_state_dict(q)
    prompt = torch.from_numpy(z["prompt"])
    got = generate(model=model, prompt=prompt, max_new_tokens=12, temperature=0.7, top_p=0.7, top_k=1, seed=1234).cpu()
    want = O.generate(O.DualAROracle(cfg, q), prompt, 12, 0.7, 0.7, 1, uniform_fn=O.FmiUniform(1234, 0))
    k = O.robust_prefix(torch.from_numpy(z["greedy_margins_ulps"]), 2.0)
    T = prompt.shape[1]
    assert torch.equal(got[:, : T + min(k, 12)], want[:, : T + min(k, 12)])


# ------------------------------------------------------------------------------- round-2 regressions


def _run_slots(model, cfg, prompts, max_new, seeds, top_k=1, extra_frames=0):
    """prefill + decode through the slot API with per-slot max_new; returns per-slot (frames, done)."""
    n = len(prompts)
    slots = list(range(n))
    samp = [model._sampling(0.7, 0.7, top_k, seeds[i], True) for i in range(n)]
    model.prefill(slots, prompts, max_new, samp)
    model.decode(slots, max(max_new) - 1 + extra_frames)
    out = [model.read(i) for i in slots]
    for i in slots:
        model.release(i)
    return out


def test_finished_slot_at_a_page_boundary_does_not_touch_other_slots():
    """A slot that ends exactly at its limit parks its position on `limit`; with limit % 64 == 0 that position's page
    was never reserved in round 1 and later frames (run because another slot of the batch is still going) appended
    K/V through block-table entry 0 -- another live utterance's first page.  Slot 1 here ends at position 128 while
    slot 0 (owner of page 0) and slot 2 keep decoding, plus extra frames after everybody finished: every slot must
    equal its standalone run."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, _ = load_dualar_case("tiny_peaky")
    model = _make_model(cfg, state, max_batch=3)
    model.set_ignore_eos(True)
    prompts = [O.make_prompt(cfg, 30, seed=21, n_semantic=6), O.make_prompt(cfg, 65, seed=22, n_semantic=9),
               O.make_prompt(cfg, 1, seed=23)]
    max_new = [150, 64, 100]        # slot 1: limit = 65 + 64 - 1 = 128
    seeds = [5, 6, 7]
    got = _run_slots(model, cfg, prompts, max_new, seeds, extra_frames=9)
    for i, p in enumerate(prompts):
        alone = generate(model=model, prompt=p, max_new_tokens=max_new[i], temperature=0.7, top_p=0.7, top_k=1,
                         seed=seeds[i], stop_on_im_end=False)
        frames, done = got[i]
        assert frames.shape[0] == max_new[i] and done == 2
        assert torch.equal(frames.t().long(), alone[:, p.shape[1]:]), f"slot {i} differs from its standalone run"


def test_out_of_range_prompt_ids_raise_like_the_reference_embedding():
    from fish_speech_amd.dual_ar import generate

    cfg, state, _ = load_dualar_case("tiny")
    model = _make_model(cfg, state)
    for row, val in ((0, cfg.vocab_size), (0, -1), (2, cfg.codebook_size), (1, -3)):
        p = O.make_prompt(cfg, 9, seed=1, n_semantic=3)
        p[row, 4] = val
        with pytest.raises(IndexError):
            generate(model=model, prompt=p, max_new_tokens=2, top_k=1, temperature=0.7, top_p=0.7)
        with pytest.raises(IndexError):
            model.forward_generate(p.view(1, cfg.num_codebooks + 1, -1).to(DEV), torch.arange(9, device=DEV))


def test_generate_batch_device_refuses_blocks_it_cannot_fill():
    from fish_speech_amd.dual_ar import generate_batch_device

    cfg, state, _ = load_dualar_case("tiny")
    model = _make_model(cfg, state, max_batch=2)
    p = [O.make_prompt(cfg, 10, seed=1)]
    with pytest.raises(ValueError):      # EOS could end a slot early: stale rows
        generate_batch_device(model=model, prompts=p, max_new_tokens=4, seeds=[1], top_k=1)
    model.set_ignore_eos(True)
    with pytest.raises(ValueError):      # beyond max_seq_len: the C side would clamp silently
        generate_batch_device(model=model, prompts=p, max_new_tokens=cfg.max_seq_len, seeds=[1], top_k=1)
    codes = generate_batch_device(model=model, prompts=p, max_new_tokens=4, seeds=[1], top_k=1)
    assert codes.shape == (1, cfg.num_codebooks, 4)


def test_sampler_variant_follows_the_live_slots():
    """top_k > 64 selects the general sampler kernel and re-captures the graphs; once that request is released a
    top_k <= 64 request must run (and give the results of) the small variant again."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, _ = load_dualar_case("mid")
    prompt = O.make_prompt(cfg, 20, seed=3, n_semantic=5)
    kw = dict(prompt=prompt, max_new_tokens=10, temperature=0.7, top_p=0.7, seed=77, stop_on_im_end=False)
    fresh = generate(model=_make_model(cfg, state), top_k=30, **kw)
    model = _make_model(cfg, state)
    big = generate(model=model, top_k=200, **kw)
    again = generate(model=model, top_k=30, **kw)
    assert torch.equal(again, fresh)
    assert big.shape == fresh.shape


def test_forward_generate_hidden_is_unnormed_without_norm_fastlayer_input():
    """llama.py:459-461: hidden_states = slow_out if norm_fastlayer_input else x (legacy dual_ar configs)."""
    import dataclasses

    cfg = O.DualARConfig(norm_fastlayer_input=False)
    state = O.make_synthetic_state(cfg, seed=4, head_gain=8.0)
    model = _make_model(cfg, state)
    orc = O.DualAROracle(cfg, state)
    orc.setup_caches(1, cfg.max_seq_len)
    p = O.make_prompt(cfg, 11, seed=2, n_semantic=4)
    ncb1 = cfg.num_codebooks + 1
    for x, pos in ((p.view(1, ncb1, -1), torch.arange(11)), (p[:, -1:].reshape(1, ncb1, 1), torch.tensor([11]))):
        _, want = orc.forward_generate(x, pos, math_backend=True)
        got = model.forward_generate(x.to(DEV), pos.to(DEV)).hidden_states.cpu()
        normed = O.rms_norm(want, state["norm.weight"], cfg.norm_eps)
        rel = float((got.float() - want.float()).norm() / want.float().norm())
        rel_n = float((got.float() - normed.float()).norm() / normed.float().norm())
        assert rel <= 2e-2 < rel_n, (rel, rel_n)


@pytest.mark.parametrize("norm_in", [True, False])
def test_fast_project_in_when_fast_dim_differs(norm_in):
    """llama.py:665-668,827: with fast_dim != dim the hidden state goes through Linear(dim, fast_dim) WITH bias before
    the fast transformer; `forward_generate` hands back the projected rows (prefill and single-token calls), the
    bias is added before the one bf16 rounding (torch's addmm), a batch of ragged prompts equals the single runs,
    and int8 + projection is refused like upstream's bias-free int8 Linear would."""
    from fish_speech_amd.dual_ar import generate, generate_batch

    kw = dict(fast_dim=96, fast_n_head=3, fast_n_local_heads=1, fast_head_dim=32, fast_intermediate_size=192)
    cfg = O.DualARConfig(norm_fastlayer_input=norm_in, **kw)
    state = O.make_synthetic_state(cfg, seed=6, head_gain=8.0)
    state["fast_project_in.bias"] = (state["fast_project_in.bias"].float() * 20).bfloat16()   # make the bias matter
    model = _make_model(cfg, state)
    orc = O.DualAROracle(cfg, state)
    orc.setup_caches(1, cfg.max_seq_len)
    p = O.make_prompt(cfg, 11, seed=2, n_semantic=4)
    ncb1 = cfg.num_codebooks + 1
    for x, pos in ((p.view(1, ncb1, -1), torch.arange(11)), (p[:, -1:].reshape(1, ncb1, 1), torch.tensor([11]))):
        _, want = orc.forward_generate(x, pos, math_backend=True)
        got = model.forward_generate(x.to(DEV), pos.to(DEV)).hidden_states.cpu()
        assert got.shape[-1] == 96 == want.shape[-1]
        nobias = want.float() - state["fast_project_in.bias"].float()
        rel = float((got.float() - want.float()).norm() / want.float().norm())
        rel_nb = float((got.float() - nobias).norm() / want.float().norm())
        assert rel <= 2e-2 < rel_nb, (rel, rel_nb)
    prompts = [O.make_prompt(cfg, T, seed=30 + T, n_semantic=4) for T in (9, 17, 5)]
    single = [generate(model=model, prompt=q, max_new_tokens=6, temperature=0.7, top_p=0.7, top_k=1, seed=3) for q in prompts]
    batch = generate_batch(model=model, prompts=prompts, max_new_tokens=6, temperature=0.7, top_p=0.7, top_k=1, seeds=[3, 3, 3])
    for a, b in zip(single, batch):
        assert torch.equal(a, b)
    from fish_speech_amd._lib import FishmiError
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR

    pc = DualARConfig.from_any(cfg, cfg.im_end_id)
    pc.weight_int8 = True
    with pytest.raises(FishmiError, match="int8"):
        MiDualAR(pc, device=DEV, im_end_id=cfg.im_end_id)


def test_decode_one_token_accepts_only_the_generate_bias():
    from fish_speech_amd.dual_ar import decode_one_token

    cfg, state, z = load_dualar_case("tiny")
    model = _make_model(cfg, state)
    prompt = torch.from_numpy(z["prompt"])
    ncb1 = cfg.num_codebooks + 1
    x, pos = prompt.view(1, ncb1, -1), torch.arange(prompt.shape[1])
    t = torch.tensor(0.7).bfloat16()
    bias = O.semantic_logit_bias(cfg, torch.bfloat16)
    import itertools

    # both calls draw the SAME uniforms: like upstream, even top_k = 1 depends on them (a draw of exactly 0 -- one in 256
    # in bf16 -- sends the exponential race to token 0), and an unseeded call takes a fresh seed
    torch.manual_seed(1234)
    model._seed_counter = itertools.count()
    a = decode_one_token(model, x, pos, t, t, 1, semantic_logit_bias=bias)
    model._seed_counter = itertools.count()
    b = decode_one_token(model, x, pos, t, t, 1)
    assert a.shape == (ncb1, 1) and torch.equal(a[1:], b[1:])
    bad = bias.clone()
    bad[0, 0, cfg.semantic_begin_id + 3] = -5.0
    with pytest.raises(NotImplementedError):
        decode_one_token(model, x, pos, t, t, 1, semantic_logit_bias=bad)


# ------------------------------------------------------------------------------- MFMA flash-attention prefill


def _long_cfg(kind, max_seq):
    from oracle.search_golden import MID

    kw = dict(MID) if kind == "mid" else {}
    kw["max_seq_len"] = max_seq
    return O.DualARConfig(**kw)


@pytest.mark.parametrize("kind,T", [("mid", 200), ("mid", 1024), ("mid", 2048), ("tiny", 333), ("tiny", 17)])
def test_mfma_flash_attention_prefill_vs_oracle_and_valu_kernel(kind, T):
    """Prefill attention on MFMA with LDS-staged K/V tiles (llama.py:910-934) at T = 200, 1024 and 2048 (head_dim 128,
    4 query heads per kv head: the S2 geometry) and at head_dim 32 / 2 heads per kv head, voice-clone shaped prompts
    (semantic tail with codes): logits and hidden state of the last position against the CPU oracle (relative L2
    <= 2 %, the bf16 noise bound of the other parity tests) and against the round-1 VALU kernel run on the same
    cache (<= 1 %: both are fp32-softmax kernels, only the summation order and the bf16 hi+lo split of the
    probabilities differ)."""
    cfg = _long_cfg(kind, 2304 if T > 400 else 512)
    state = O.make_synthetic_state(cfg, seed=5, head_gain=8.0)
    model = _make_model(cfg, state, max_batch=1)
    orc = O.DualAROracle(cfg, state)
    orc.setup_caches(1, cfg.max_seq_len)
    p = O.make_prompt(cfg, T, seed=T, n_semantic=T // 2)
    ncb1 = cfg.num_codebooks + 1
    x, pos = p.view(1, ncb1, -1), torch.arange(T)
    want_logits, want_hidden = orc.forward_generate(x, pos, math_backend=True)
    ids = model._table(1, torch.int32).view(-1).long().cpu()
    out = {}
    for impl in (1, 0):
        model.set_attn_impl(impl)
        r = model.forward_generate(x.to(DEV), pos.to(DEV))
        out[impl] = (r.logits[0, 0].float().cpu()[ids], r.hidden_states.float().cpu().view(-1))
    model.set_attn_impl(1)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    wl, wh = want_logits[0, 0].float()[ids], want_hidden.float().view(-1)
    e = dict(mfma_vs_oracle=(rel(out[1][0], wl), rel(out[1][1], wh)), valu_vs_oracle=(rel(out[0][0], wl), rel(out[0][1], wh)),
             mfma_vs_valu=(rel(out[1][0], out[0][0]), rel(out[1][1], out[0][1])))
    print(kind, T, e)
    assert max(e["mfma_vs_oracle"]) <= 2e-2, e
    assert max(e["mfma_vs_valu"]) <= 1e-2, e
    assert max(e["mfma_vs_oracle"]) <= 1.5 * max(e["valu_vs_oracle"]) + 4e-3, e   # a bf16 step or two of either


def test_mfma_flash_attention_ragged_batch_equals_single_prompts():
    """Query tiles of different utterances (lengths 1, 16, 17, 130, 47: full tiles, one-row tiles, padded tiles) in
    one prefill: every utterance's prefill-frame logits equal those of its own batch-1 prefill bit for bit (a tile
    only ever reads its own slot's pages), and the first frame's tokens agree."""
    cfg, state, _ = load_dualar_case("mid_peaky")
    model = _make_model(cfg, state, max_batch=5)
    lens = [1, 16, 17, 130, 47]
    prompts = [O.make_prompt(cfg, T, seed=40 + i, n_semantic=T // 3) for i, T in enumerate(lens)]
    samp = [model._sampling(0.7, 0.7, 1, 9, True) for _ in lens]
    model.prefill(list(range(5)), prompts, [4] * 5, samp)
    logits, _, hidden, _ = model.debug_taps(5)
    for i in range(5):
        model.release(i)
    for i, p in enumerate(prompts):
        model.prefill([0], [p], [4], [samp[i]])
        l1, _, h1, _ = model.debug_taps(1)
        model.release(0)
        assert torch.equal(l1[0], logits[i]) and torch.equal(h1[0], hidden[i]), i


@pytest.mark.parametrize("T", [520, 1100, 2047])
def test_mfma_decode_attention_long_context_vs_oracle_and_valu_kernel(T):
    """Decode attention at long contexts runs on MFMA (rows at or beyond a position threshold -- 1024 by default, 512 here: all query heads of a kv head in
    one work-group, key ranges split over four work-groups, partial softmax states merged, the new token folded in by
    the merge kernel).  After a T-token voice-clone shaped prompt (head_dim 128, 4 query heads per kv head: the S2
    geometry) three teacher-forced decode positions: logits and hidden state against the CPU oracle (relative L2
    <= 2 %) and against the fused VALU kernel on the same cache (<= 1 %), which also proves that the K/V rows the
    merge kernel appends are the ones the VALU kernel would have appended (the later positions read them)."""
    cfg = _long_cfg("mid", 2304)
    state = O.make_synthetic_state(cfg, seed=6, head_gain=8.0)
    model = _make_model(cfg, state, max_batch=1)
    orc = O.DualAROracle(cfg, state)
    orc.setup_caches(1, cfg.max_seq_len)
    p = O.make_prompt(cfg, T, seed=T + 1, n_semantic=T // 2)
    ncb1 = cfg.num_codebooks + 1
    g = torch.Generator().manual_seed(T)
    frames = torch.zeros(3, ncb1, dtype=torch.int64)
    frames[:, 1:] = torch.randint(0, cfg.codebook_size, (3, cfg.num_codebooks), generator=g)
    frames[:, 0] = frames[:, 1] + cfg.semantic_begin_id
    ids = model._table(1, torch.int32).view(-1).long().cpu()
    orc.forward_generate(p.view(1, ncb1, -1), torch.arange(T), math_backend=True)
    want = [orc.forward_generate(frames[i].view(1, ncb1, 1), torch.tensor([T + i]), math_backend=True) for i in range(3)]
    rel = lambda a, b: float((a - b).norm() / b.norm())
    out = {}
    for thr in (512, 0):
        model.set_attn_long_threshold(thr)
        model.forward_generate(p.view(1, ncb1, -1).to(DEV), torch.arange(T, device=DEV))
        res = []
        for i in range(3):
            r = model.forward_generate(frames[i].view(1, ncb1, 1).to(DEV), torch.tensor([T + i], device=DEV))
            res.append((r.logits[0, 0].float().cpu()[ids], r.hidden_states.float().cpu().view(-1)))
        out[thr] = res
    model.set_attn_long_threshold(1024)
    for i in range(3):
        wl, wh = want[i][0][0, 0].float()[ids], want[i][1].float().view(-1)
        e = dict(mfma_vs_oracle=(rel(out[512][i][0], wl), rel(out[512][i][1], wh)),
                 valu_vs_oracle=(rel(out[0][i][0], wl), rel(out[0][i][1], wh)),
                 mfma_vs_valu=(rel(out[512][i][0], out[0][i][0]), rel(out[512][i][1], out[0][i][1])))
        print("decode position", T + i, e)
        assert max(e["mfma_vs_oracle"]) <= 2e-2, e
        assert max(e["mfma_vs_valu"]) <= 1e-2, e
        assert max(e["mfma_vs_oracle"]) <= 1.5 * max(e["valu_vs_oracle"]) + 4e-3, e


def test_decode_attention_regimes_are_batch_invariant_and_graph_equals_eager():
    """Which decode-attention kernel serves a row depends only on that row's position (threshold 512), so an utterance
    keeps its batch-1 result whatever its batch-mates' contexts are: a ragged batch with prompts of 40, 700, 505 and
    1300 tokens (short, long, one CROSSING the threshold during the run, long) -- the frame graph then launches both
    kernels -- equals the four batch-1 runs bit for bit, through the hipGraph and eagerly; and the well-conditioned
    fixture's greedy tokens do not change when every row is forced through the MFMA path."""
    from fish_speech_amd.dual_ar import generate, generate_batch

    cfg = _long_cfg("mid", 2304)
    state = O.make_peaky_state(cfg, seed=93, emb_gain=2.5, slow_gain=3.0, fast_gain=2.5)
    model = _make_model(cfg, state, max_batch=4)
    model.set_attn_long_threshold(512)
    lens = [40, 700, 505, 1300]
    prompts = [O.make_prompt(cfg, T, seed=60 + i, n_semantic=T // 3) for i, T in enumerate(lens)]
    kw = dict(max_new_tokens=14, temperature=0.7, top_p=0.7, top_k=30, stop_on_im_end=False)
    seeds = [31, 32, 33, 34]
    model.set_graph(True)
    batch = generate_batch(model=model, prompts=prompts, seeds=seeds, **kw)
    model.set_graph(False)
    eager = generate_batch(model=model, prompts=prompts, seeds=seeds, **kw)
    model.set_graph(True)
    for i, p in enumerate(prompts):
        assert torch.equal(batch[i], eager[i]), f"graph != eager for utterance {i}"
        single = generate_batch(model=model, prompts=[p], seeds=[seeds[i]], **kw)[0]
        assert torch.equal(single, batch[i]), f"utterance {i} (prompt {lens[i]}) differs from its batch-1 run"
    # the decisions of a well-conditioned model do not depend on the attention kernel (threshold 1: MFMA for every row)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dualar_mid_peaky.npz"))
    prompt = torch.from_numpy(z["prompt"])
    gk = dict(prompt=prompt, max_new_tokens=int(z["max_new"]), temperature=0.7, top_p=0.7, top_k=1, seed=int(z["uniform_seed"]))
    model.set_attn_long_threshold(0)
    valu = generate(model=model, **gk)
    model.set_attn_long_threshold(32)
    mfma = generate(model=model, **gk)
    model.set_attn_long_threshold(1024)
    assert torch.equal(valu, mfma) and np.array_equal(valu.numpy(), z["tokens"])


def test_prefix_reuse_is_bit_identical_for_every_length_class():
    """MiDualAR.prefill(reuse_prefix=True) against plain prefills, slot kept between calls: a long prompt extended
    (suffix of 3 rows: forced through the tiled GEMM), a short prompt (<= 16 rows: decode GEMV) extended to a long one
    (no reuse: the kernels differ), a short one extended within 16 rows, an unrelated prompt (no common prefix), and
    the same prompt again (all but the last column reused).  Every generation equals the one without reuse."""
    from fish_speech_amd.dual_ar import generate

    cfg, state, _ = load_dualar_case("mid_peaky")
    model = _make_model(cfg, state, max_batch=1)
    base = O.make_prompt(cfg, 70, seed=3, n_semantic=20)
    other = O.make_prompt(cfg, 33, seed=4, n_semantic=5)
    seq = [base[:, :40], base[:, :43], base[:, :70], other, base[:, :9], base[:, :14], base[:, :30], base[:, :30]]
    kw = dict(max_new_tokens=6, temperature=0.7, top_p=0.7, top_k=1, seed=11)
    want = [generate(model=model, prompt=p, **kw) for p in seq]
    model.prefilled_rows = model.reused_rows = 0
    got = [generate(model=model, prompt=p, reuse_prefix=True, **kw) for p in seq]
    model.release(0)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i
    # 40 | +3 (40 reused) | +27 (43 reused) | 33 new | 9 new | +5 (9 reused, both short) | 30 new (short -> long) | 29 reused
    assert model.reused_rows == 40 + 43 + 9 + 29, model.reused_rows
    assert model.prefilled_rows == sum(p.shape[1] for p in seq) - model.reused_rows


def test_prefill_attention_wide_query_tiles_equal_the_16_row_tiles_bit_for_bit():
    """Round 4: from 2048 / 4096 rows per prefill call the MFMA prefill attention gives a work-group 32 / 48 query rows
    (two / three column groups sharing the staged K/V blocks) instead of 16.  A query row's arithmetic does not depend
    on the group it sits in: the first-frame logits and hidden rows of every utterance of a 9-prompt call (4.3 k rows:
    48-row tiles), of its first 5 prompts (2.4 k rows: 32-row tiles) and of each prompt alone (16-row tiles) are equal
    bit for bit -- which also pins the 256-column prefill GEMM's tile heights (chosen from the row count) against each other."""
    cfg, state, _ = load_dualar_case("mid_peaky")
    model = _make_model(cfg, state, max_batch=9)
    prompts = [O.make_prompt(cfg, 500 - 7 * i, seed=50 + i, n_semantic=100) for i in range(9)]
    assert sum(p.shape[1] for p in prompts) >= 4096 and 2048 <= sum(p.shape[1] for p in prompts[:5]) < 4096

    def first_frame(idx):
        sp = [model._sampling(0.7, 0.7, 1, 100 + i, True) for i in idx]
        model.prefill(list(range(len(idx))), [prompts[i] for i in idx], [2] * len(idx), sp)
        logits, _, hidden, _ = model.debug_taps(len(idx))
        for s in range(len(idx)):
            model.release(s)
        return logits.cpu(), hidden.cpu()

    l9, h9 = first_frame(list(range(9)))
    l5, h5 = first_frame(list(range(5)))
    assert torch.equal(l9[:5], l5) and torch.equal(h9[:5], h5)
    for i in (0, 4, 8):
        l1, h1 = first_frame([i])
        assert torch.equal(l9[i : i + 1], l1) and torch.equal(h9[i : i + 1], h1), i
    assert bool(torch.isfinite(l9[:, :8].float()).all())

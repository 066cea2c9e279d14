"""Streaming chunked decode (BASELINE.json config 5) on the GPU: what is streamed must be exactly what
the offline path produces -- codes from ``generate_batch``, audio from ``MiDAC.from_indices`` over the
final codes (``codes = y[1:, T:-1]``, inference.py:708) -- for any chunking."""
import time

import pytest
import torch

from oracle import dac as D
from tests.helpers import load_dualar_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def small_codec():
    from fish_speech_amd.dac import DacConfig, MiDAC

    cfg = D.small_config()
    state = D.make_synthetic_state(cfg, seed=11)
    return cfg, state, MiDAC.from_state_dict(DacConfig.from_any(cfg), state, device=DEV)


@pytest.fixture(scope="module")
def full_codec():
    from fish_speech_amd.dac import DacConfig, MiDAC

    cfg = D.DacConfig()
    state = D.make_synthetic_state(cfg, seed=3)
    return cfg, state, MiDAC.from_state_dict(DacConfig.from_any(cfg), state, device=DEV)


def _tail_chunks_equal_full(cfg, codec, B, T, cuts, seed):
    codes = D.make_codes(cfg, B, T, seed=seed).to(DEV)
    want = codec.from_indices(codes.clone())
    fl = cfg.frame_length
    pieces, t0 = [], 0
    for t1 in list(cuts) + [T]:
        got = codec.from_indices_tail(codes[:, :, :t1].clone(), t0)
        assert got.shape == (B, 1, (t1 - t0) * fl)
        pieces.append(got)
        t0 = t1
    got = torch.cat(pieces, dim=-1)
    assert torch.equal(got, want), float((got - want).abs().max())


def test_tail_decode_concatenates_to_the_full_decode_bitwise_small(small_codec):
    cfg, _, codec = small_codec
    assert codec.context_frames >= 1
    _tail_chunks_equal_full(cfg, codec, 2, 40, [1, 2, 7, 8, 20, 39], seed=1)
    _tail_chunks_equal_full(cfg, codec, 1, 9, [4], seed=2)


def test_tail_decode_concatenates_to_the_full_decode_bitwise_full_size(full_codec):
    """yaml-sized codec; cuts closer together than the decoder's receptive field (5 frames) and far apart."""
    cfg, _, codec = full_codec
    assert codec.context_frames == 5   # 19 latent columns for decoder rates (8, 8, 4, 2)
    _tail_chunks_equal_full(cfg, codec, 2, 48, [3, 4, 9, 33], seed=4)


def _cached_chunks_equal_full(cfg, codec, B, T, cuts, seed, sid):
    codes = D.make_codes(cfg, B, T, seed=seed).to(DEV)
    want = codec.from_indices(codes.clone())
    pieces, t0 = [], 0
    for t1 in list(cuts) + [T]:
        pieces.append(codec.from_indices_tail(codes[:, :, :t1].clone(), t0, stream_id=sid))
        t0 = t1
    got = torch.cat(pieces, dim=-1)
    assert torch.equal(got, want), float((got - want).abs().max())


def test_cached_tail_decode_is_bit_identical_and_survives_interleaving(small_codec, full_codec):
    """from_indices_tail(stream_id=...): the quantizer side of frames [0, t0) comes from the state of the previous call
    (per-layer K/V of the windowed transformer, its output, the upsampled latents).  Chunks shorter and longer than
    the upsampler's (5 frames) and the transformer's (window) context; a second stream with the SAME chunk marks
    interleaved on the same codec (its calls restart the state: correct, only slower); a gap in t0; a batch change."""
    for cfg, _, codec in (small_codec, full_codec):
        sid = codec.new_stream_id()
        _cached_chunks_equal_full(cfg, codec, 2, 48, [1, 2, 7, 8, 20, 47], seed=5, sid=sid)
        _cached_chunks_equal_full(cfg, codec, 2, 30, [9], seed=6, sid=codec.new_stream_id())
        # two interleaved streams, same marks
        a = D.make_codes(cfg, 2, 40, seed=7).to(DEV)
        b = D.make_codes(cfg, 2, 40, seed=8).to(DEV)
        wa, wb = codec.from_indices(a.clone()), codec.from_indices(b.clone())
        ia, ib = codec.new_stream_id(), codec.new_stream_id()
        pa, pb, t0 = [], [], 0
        for t1 in (9, 24, 40):
            pa.append(codec.from_indices_tail(a[:, :, :t1].clone(), t0, stream_id=ia))
            pb.append(codec.from_indices_tail(b[:, :, :t1].clone(), t0, stream_id=ib))
            t0 = t1
        assert torch.equal(torch.cat(pa, -1), wa) and torch.equal(torch.cat(pb, -1), wb)
        # a call that does not continue the state (gap), then one with another batch size
        fl = cfg.frame_length
        got = codec.from_indices_tail(a[:, :, :33].clone(), 30, stream_id=ia)
        assert torch.equal(got, wa[..., 30 * fl:33 * fl])
        got = codec.from_indices_tail(a[:1, :, :36].clone(), 33, stream_id=ia)
        assert torch.equal(got, codec.from_indices(a[:1].clone())[..., 33 * fl:36 * fl])
        codec.stream_reset()
        got = codec.from_indices_tail(a[:, :, :12].clone(), 3, stream_id=ia)
        assert torch.equal(got, wa[..., 3 * fl:12 * fl])
        # round 6: the decoder's kept left context.  A call that had to START OVER at t0 > 0 (no state: the receptive
        # field is recomputed from zero tails, its audio discarded) leaves exact tails behind, so the calls that CONTINUE
        # it -- one frame, then more than the longest tail (54 columns = 2 frames at the first block) -- are offline's bits
        ic = codec.new_stream_id()
        got = codec.from_indices_tail(b[:, :, :20].clone(), 14, stream_id=ic)
        assert torch.equal(got, wb[..., 14 * fl:20 * fl])
        got = codec.from_indices_tail(b[:, :, :21].clone(), 20, stream_id=ic)
        assert torch.equal(got, wb[..., 20 * fl:21 * fl])
        got = codec.from_indices_tail(b[:, :, :40].clone(), 21, stream_id=ic)
        assert torch.equal(got, wb[..., 21 * fl:40 * fl])
    codec.stream_reset()


@pytest.mark.parametrize("mode", ["fp16_split", "autocast_bf16"])
def test_streamed_chunks_equal_offline_for_random_schedules(full_codec, mode):
    """Round 6 (decoder tails kept per stream): random chunk schedules -- one-frame chunks, chunks shorter and longer than
    the longest kept tail (54 columns = 2 frames at the first decoder block), batch 1 and 3 -- concatenate to the offline
    decode bit for bit, in the default fp16-split arithmetic and in the engine's autocast(bf16) mode (one operand plane:
    the tails then hold one plane per column)."""
    import contextlib
    import random

    cfg, _, codec = full_codec
    ctx = (lambda: torch.autocast(device_type="cuda", dtype=torch.bfloat16)) if mode == "autocast_bf16" else contextlib.nullcontext
    rng = random.Random(606)
    for trial in range(4):
        B = 1 if trial % 2 == 0 else 3
        T = rng.randint(20, 70)
        cuts = sorted(set(rng.sample(range(1, T), rng.randint(2, 7))))
        codes = D.make_codes(cfg, B, T, seed=100 + trial).to(DEV)
        with ctx():
            want = codec.from_indices(codes.clone())
            sid = codec.new_stream_id()
            pieces, t0 = [], 0
            for t1 in cuts + [T]:
                pieces.append(codec.from_indices_tail(codes[:, :, :t1].clone(), t0, stream_id=sid))
                t0 = t1
            codec.close_stream(sid)
        got = torch.cat(pieces, dim=-1)
        assert got.dtype == want.dtype and torch.equal(got, want), (mode, trial, B, T, cuts, float((got.float() - want.float()).abs().max()))
    codec.stream_reset()


def test_cached_tail_decode_long_stream_full_size(full_codec):
    """beyond the attention window (tf_window frames) and across a capacity doubling (1024 frames)"""
    cfg, _, codec = full_codec
    T = 1100
    codes = D.make_codes(cfg, 1, T, seed=9).to(DEV)
    want = codec.from_indices(codes.clone())
    sid = codec.new_stream_id()
    fl = cfg.frame_length
    t0 = 0
    for t1 in (8, 40, 200, 500, 1000, 1030, T):
        got = codec.from_indices_tail(codes[:, :, :t1].clone(), t0, stream_id=sid)
        assert torch.equal(got, want[..., t0 * fl:t1 * fl]), (t0, t1)
        t0 = t1
    codec.stream_reset()


def test_tail_decode_matches_oracle(full_codec):
    cfg, state, codec = full_codec
    codes = D.make_codes(cfg, 1, 12, seed=8)
    want = D.DacOracle(cfg, state).from_indices(codes.clone())
    got = codec.from_indices_tail(codes.to(DEV), 7).cpu()
    ref = want[..., 7 * cfg.frame_length:]
    assert float((got - ref).pow(2).mean().sqrt()) <= 1e-4


def test_tail_rejects_bad_ranges(small_codec):
    cfg, _, codec = small_codec
    codes = D.make_codes(cfg, 1, 6, seed=3).to(DEV)
    for t0 in (-1, 6, 9):
        with pytest.raises((ValueError, RuntimeError)):
            codec.from_indices_tail(codes.clone(), t0)


@pytest.fixture(scope="module")
def tiny_model():
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR

    cfg, state, z = load_dualar_case("tiny")
    model = MiDualAR.from_state_dict(DualARConfig.from_any(cfg), state, device=DEV, im_end_id=cfg.im_end_id)
    model.setup_caches(4, cfg.max_seq_len)
    return cfg, model


def _prompts(cfg, n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        T = 5 + 3 * i
        p = torch.zeros(cfg.num_codebooks + 1, T, dtype=torch.int64)
        p[0] = torch.randint(0, cfg.semantic_begin_id, (T,), generator=g)
        out.append(p)
    return out


@pytest.mark.parametrize("ignore_eos", [True, False])
@pytest.mark.parametrize("first,chunk", [(1, 1), (3, 5), (8, 32)])
def test_stream_equals_offline_codes_and_audio(tiny_model, small_codec, ignore_eos, first, chunk):
    """Sampled generation (temperature 0.9, top-p 0.8, top-k 20, RAS on) of a ragged batch of three, streamed
    vs. offline: identical codes, bit-identical audio, per utterance, whatever the chunk sizes.  With EOS live
    the utterances end at different frames (or at max_new_tokens)."""
    from fish_speech_amd.dual_ar import generate_batch
    from fish_speech_amd.stream import generate_stream

    cfg, model = tiny_model
    ccfg, _, codec = small_codec
    assert cfg.num_codebooks == ccfg.n_codebooks + 1
    model.set_ignore_eos(ignore_eos)
    prompts = _prompts(cfg, 3, seed=5)
    kw = dict(temperature=0.9, top_p=0.8, top_k=20, seeds=[101, 102, 103])
    n_new = 24
    offline = generate_batch(model=model, prompts=prompts, max_new_tokens=n_new, **kw)
    audio = [[] for _ in prompts]
    codes = [[] for _ in prompts]
    marks = []
    for ch in generate_stream(model=model, codec=codec, prompts=prompts, max_new_tokens=n_new,
                              first_chunk_frames=first, chunk_frames=chunk, **kw):
        marks.append((ch.t0, ch.t1))
        assert ch.audio.shape[-1] == (ch.t1 - ch.t0) * ccfg.frame_length
        for i, v in enumerate(ch.valid_frames):
            if v:
                audio[i].append(ch.audio[i, :, : v * ccfg.frame_length])
                codes[i].append(ch.codes[i, :, :v])
    assert marks[0] == (0, min(first, n_new - 1)) or not ignore_eos
    assert all(a[1] == b[0] for a, b in zip(marks, marks[1:]))
    lengths = []
    for i, p in enumerate(prompts):
        want_codes = offline[i][1:, p.shape[1]:-1].to(DEV)       # inference.py:708
        lengths.append(want_codes.shape[1])
        got_codes = torch.cat(codes[i], dim=1) if codes[i] else want_codes[:, :0]
        assert torch.equal(got_codes, want_codes), i
        if want_codes.shape[1] == 0:
            assert not audio[i]
            continue
        want_audio = codec.from_indices(want_codes[None].clone())[0]
        assert torch.equal(torch.cat(audio[i], dim=-1), want_audio), i
    if ignore_eos:
        assert lengths == [n_new - 1] * 3
    model.set_ignore_eos(False)


def test_generate_long_end_to_end_on_the_gpu():
    """Text -> prompt builder -> real Dual-AR kernels -> codes, two text chunks with context carry
    (SURVEY.md row a15).  Each chunk's codes must equal a direct `generate` on the prompt generate_long built
    (same seed), and the second prompt must contain the first chunk's codes as VQ columns."""
    from fish_speech_amd import text2semantic as T2S
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate
    from oracle import dual_ar as O
    from oracle.fake_tokenizer import ByteTokenizer

    tok = ByteTokenizer()
    cfg = O.DualARConfig(vocab_size=tok.vocab_size + 4, dim=128, n_layer=2, n_head=4, n_local_heads=2, head_dim=32,
                         intermediate_size=256, max_seq_len=2048 + 512, codebook_size=4096, num_codebooks=10,
                         semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                         im_end_id=tok.get_token_id("<|im_end|>"), n_fast_layer=2)
    state = O.make_synthetic_state(cfg, seed=5, head_gain=4.0)
    model = MiDualAR.from_state_dict(DualARConfig.from_any(cfg), state, device=DEV, im_end_id=cfg.im_end_id)
    model.tokenizer = tok
    model.setup_caches(1, cfg.max_seq_len)
    prompts = []

    def recording_generate(**kw):
        prompts.append(kw["prompt"].cpu().clone())
        kw.pop("decode_one_token", None), kw.pop("audio_masks", None), kw.pop("audio_parts", None)
        return generate(seed=1234 + len(prompts), **kw)

    T2S.generate = recording_generate
    try:
        text = "<|speaker:0|>First chunk of text.<|speaker:1|>Second chunk, another speaker."
        out = list(T2S.generate_long(model=model, device=DEV, text=text, max_new_tokens=12, chunk_length=30,
                                     temperature=0.8, top_p=0.8, top_k=20))
    finally:
        T2S.generate = T2S._default_generate
    assert [r.action for r in out] == ["sample", "sample", "next"] and len(prompts) == 2
    for i, r in enumerate(out[:2]):
        assert r.codes.shape[0] == 10 and 1 <= r.codes.shape[1] <= 11
        assert int(r.codes.min()) >= 0 and int(r.codes.max()) < 4096
        y = generate(model=model, prompt=prompts[i].to(DEV), max_new_tokens=12, seed=1235 + i, temperature=0.8,
                     top_p=0.8, top_k=20)
        assert torch.equal(y[1:, prompts[i].shape[1]:-1].cpu(), r.codes.cpu())
    n0 = out[0].codes.shape[1]
    p1 = prompts[1]
    vq = (p1[0] >= tok.semantic_begin_id) & (p1[0] <= tok.semantic_end_id)
    assert int(vq.sum()) == n0 and torch.equal(p1[1:, vq], out[0].codes.cpu())
    assert torch.equal(p1[:, : prompts[0].shape[1]], prompts[0])          # the first prompt is a prefix of the second


def test_generate_long_prefix_kv_reuse_is_bit_identical_to_re_prefill():
    """8f #1: every chunk of generate_long repeats the conversation so far; upstream re-prefills all of it
    (inference.py:620-688).  With prefix-KV reuse the slot keeps its pages and only the new columns are prefilled.
    A 3-chunk conversation with a voice-clone system prompt, reuse on vs off: identical codes for every chunk
    (bit-identical K/V: the reused positions were written by a prefill, the suffix rows run through the same
    kernels a full prefill uses), and most prompt rows are not recomputed."""
    import itertools
    import time

    from fish_speech_amd import text2semantic as T2S
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR
    from oracle import dual_ar as O
    from oracle.fake_tokenizer import ByteTokenizer

    tok = ByteTokenizer()
    cfg = O.DualARConfig(vocab_size=tok.vocab_size + 4, dim=128, n_layer=2, n_head=4, n_local_heads=2, head_dim=32,
                         intermediate_size=256, max_seq_len=2048 + 1024, codebook_size=4096, num_codebooks=10,
                         semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                         im_end_id=tok.get_token_id("<|im_end|>"), n_fast_layer=2)
    state = O.make_synthetic_state(cfg, seed=5, head_gain=4.0)
    model = MiDualAR.from_state_dict(DualARConfig.from_any(cfg), state, device=DEV, im_end_id=cfg.im_end_id)
    model.tokenizer = tok
    model.setup_caches(1, cfg.max_seq_len)
    ref_codes = torch.randint(0, 4096, (10, 120), generator=torch.Generator().manual_seed(3))
    text = ("<|speaker:0|>First chunk of text, long enough.<|speaker:1|>Second chunk, another speaker talks."
            "<|speaker:0|>And a third turn to finish the conversation.")
    kw = dict(model=model, device=DEV, text=text, max_new_tokens=14, chunk_length=40, temperature=0.8, top_p=0.8,
              top_k=20, prompt_text=["reference transcript"], prompt_tokens=[ref_codes])

    def run(reuse):
        torch.manual_seed(7)
        model._seed_counter = itertools.count()
        model.prefilled_rows = model.reused_rows = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = list(T2S.generate_long(reuse_prefix_kv=reuse, **kw))
        torch.cuda.synchronize()
        return out, model.prefilled_rows, model.reused_rows, time.perf_counter() - t0

    plain, rows_plain, reused_plain, t_plain = run(False)
    reuse, rows_reuse, reused, t_reuse = run(True)
    assert [r.action for r in plain] == ["sample"] * 3 + ["next"] == [r.action for r in reuse]
    for a, b in zip(plain[:3], reuse[:3]):
        assert torch.equal(a.codes.cpu(), b.codes.cpu())
    print(f"prefix-KV reuse over 3 chunks: prompt rows prefilled {rows_plain} -> {rows_reuse} ({reused} reused); "
          f"wall {t_plain * 1e3:.1f} -> {t_reuse * 1e3:.1f} ms")
    assert reused_plain == 0 and reused > 0.5 * rows_plain and rows_reuse + reused == rows_plain


def test_from_pretrained_reads_an_s2_style_checkpoint_directory(tmp_path, monkeypatch):
    """SURVEY.md row a17 end to end: config.json of model_type fish_qwen3_omni, sharded safetensors with the HF
    tensor names (text_model.model.* / audio_decoder.*) and separate wq/wk/wv (Attention.load_hook,
    llama.py:877-882), ids injected from the tokenizer -> the loaded model generates exactly what a model built
    straight from the state dict generates."""
    import json
    import sys
    import types

    from safetensors.torch import save_file

    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate
    from oracle import dual_ar as O
    from oracle.fake_tokenizer import ByteTokenizer

    tok = ByteTokenizer()
    cfg = O.DualARConfig(vocab_size=tok.vocab_size + 4, dim=128, n_layer=2, n_head=4, n_local_heads=2, head_dim=32,
                         intermediate_size=256, max_seq_len=256, codebook_size=4096, num_codebooks=10,
                         semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                         im_end_id=tok.get_token_id("<|im_end|>"), n_fast_layer=2, rope_base=1000000.0, norm_eps=1e-6)
    state = O.make_synthetic_state(cfg, seed=21, head_gain=4.0)
    hf = {}
    for k, v in state.items():
        if k.startswith("fast_"):
            name = "audio_decoder." + k[len("fast_"):]
        elif k.startswith("codebook_embeddings."):
            name = "audio_decoder." + k
        else:
            name = "text_model.model." + k
        if name.endswith("attention.wqkv.weight") and ".layers.0." in name and name.startswith("text_model"):
            q = cfg.n_head * cfg.head_dim
            kv = cfg.n_local_heads * cfg.head_dim
            pre = name[: -len("wqkv.weight")]
            hf[pre + "wq.weight"], hf[pre + "wk.weight"], hf[pre + "wv.weight"] = (
                v[:q].contiguous(), v[q:q + kv].contiguous(), v[q + kv:].contiguous())
        else:
            hf[name] = v.contiguous()
    names = sorted(hf)
    shards = {"model-00001-of-00002.safetensors": names[: len(names) // 2],
              "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for fn, ks in shards.items():
        save_file({k: hf[k] for k in ks}, str(tmp_path / fn))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps(
        {"weight_map": {k: fn for fn, ks in shards.items() for k in ks}}))
    (tmp_path / "config.json").write_text(json.dumps({
        "model_type": "fish_qwen3_omni", "semantic_start_token_id": 0, "semantic_end_token_id": 0,
        "text_config": {"vocab_size": cfg.vocab_size, "n_layer": 2, "n_head": 4, "n_local_heads": 2, "head_dim": 32,
                        "dim": 128, "intermediate_size": 256, "rope_base": 1000000, "norm_eps": 1e-6,
                        "max_seq_len": 256, "attention_qk_norm": True},
        "audio_decoder_config": {"vocab_size": 4096, "num_codebooks": 10, "n_layer": 2}}))

    class FishTokenizer:                       # stands in for fish_speech.tokenizer.FishTokenizer (no tokenizer files here)
        @classmethod
        def from_pretrained(cls, path):
            assert str(path) == str(tmp_path)
            return tok

    pkg, mod = types.ModuleType("fish_speech"), types.ModuleType("fish_speech.tokenizer")
    mod.FishTokenizer = FishTokenizer
    pkg.tokenizer = mod
    monkeypatch.setitem(sys.modules, "fish_speech", pkg)
    monkeypatch.setitem(sys.modules, "fish_speech.tokenizer", mod)

    loaded = MiDualAR.from_pretrained(str(tmp_path), device=DEV)
    assert loaded.tokenizer is tok and loaded.config.semantic_begin_id == tok.semantic_begin_id
    assert loaded.config.im_end_id == cfg.im_end_id and loaded.config.fast_attention_qk_norm is True
    direct = MiDualAR.from_state_dict(DualARConfig.from_any(cfg), state, device=DEV, im_end_id=cfg.im_end_id)
    prompt = _prompts(cfg, 1, seed=3)[0]
    kw = dict(max_new_tokens=10, temperature=0.8, top_p=0.8, top_k=20, seed=99)
    assert torch.equal(generate(model=loaded, prompt=prompt, **kw), generate(model=direct, prompt=prompt, **kw))


def test_continuous_batching_queue_is_schedule_invariant(tiny_model):
    """Seven ragged requests through three slots with refill (config 4's mixed-length serving): every utterance
    equals its standalone `generate` with the same seed, whatever the admission order or the slot it landed in."""
    from fish_speech_amd.dual_ar import generate
    from fish_speech_amd.scheduler import generate_queue, lpt_order

    cfg, model = tiny_model
    model.set_ignore_eos(False)
    prompts = _prompts(cfg, 7, seed=31)
    seeds = [900 + i for i in range(7)]
    kw = dict(max_new_tokens=30, temperature=0.9, top_p=0.8, top_k=20)
    alone = [generate(model=model, prompt=p, seed=s, **kw) for p, s in zip(prompts, seeds)]
    stats = {}
    fifo = generate_queue(model=model, prompts=prompts, seeds=seeds, max_batch=3, poll_every=4, stats=stats, **kw)
    lpt = generate_queue(model=model, prompts=prompts, seeds=seeds, max_batch=3, poll_every=7,
                         order=lpt_order([p.shape[1] for p in prompts]), **kw)
    one = generate_queue(model=model, prompts=prompts, seeds=seeds, max_batch=1, poll_every=16, **kw)
    for i in range(7):
        assert torch.equal(fifo[i], alone[i]), i
        assert torch.equal(lpt[i], alone[i]), i
        assert torch.equal(one[i], alone[i]), i
    lengths = {int(a.shape[1] - p.shape[1]) for a, p in zip(alone, prompts)}
    assert stats["frames_run"] >= max(lengths) - 1


def test_serve_stream_streams_and_refills_and_equals_the_offline_path(tiny_model, small_codec):
    """serving.serve_stream (configs 4 + 5 together): nine ragged requests through three slots, each on its own chunk
    schedule, slots refilled as utterances end by <|im_end|> or max_new_tokens.  Per utterance the concatenated
    segments are exactly `generate` + `from_indices` of that utterance alone (codes equal, audio bit-identical) --
    whatever slot, neighbours and chunk boundaries it got; 20 interleaved codec streams (more than the library parks)
    stay bit-identical too."""
    from fish_speech_amd.dual_ar import generate
    from fish_speech_amd.serving import StreamRequest, collect, serve_stream

    cfg, model = tiny_model
    ccfg, _, codec = small_codec
    model.set_ignore_eos(False)
    prompts = _prompts(cfg, 9, seed=77)
    seeds = [500 + i for i in range(9)]
    limits = [30, 12, 30, 40, 5, 30, 2, 30, 1]
    kw = dict(temperature=0.9, top_p=0.8, top_k=20)
    alone = [generate(model=model, prompt=p, seed=s, max_new_tokens=m, **kw) for p, s, m in zip(prompts, seeds, limits)]
    for step, first, chunk in ((8, 8, 32), (1, 1, 1), (5, 2, 7)):
        reqs = [StreamRequest(prompt=p, max_new_tokens=m, seed=s, rid=i) for i, (p, s, m) in enumerate(zip(prompts, seeds, limits))]
        evs = list(serve_stream(model=model, codec=codec, requests=reqs, max_batch=3, step_frames=step,
                                first_chunk_frames=first, chunk_frames=chunk, chunk_growth=1.5, **kw))
        got = collect(evs, ccfg.frame_length)
        assert sorted(e.rid for e in evs if e.kind == "final") == list(range(9))
        for i, p in enumerate(prompts):
            want = alone[i][1:, p.shape[1]:-1]                                  # inference.py:708
            if want.shape[1] == 0:
                assert i not in got or got[i][1] is None or got[i][1].shape[1] == 0
                continue
            audio, codes = got[i]
            assert torch.equal(codes, want), (step, i)
            assert torch.equal(audio, codec.from_indices(want[None].to(DEV))[0, 0].cpu()), (step, i)
    # more open codec streams than the library keeps parked (15): the least recently used restart, results unchanged
    codes = [D.make_codes(ccfg, 1, 24, seed=40 + i).to(DEV) for i in range(20)]
    full = [codec.from_indices(c.clone()) for c in codes]
    ids = [codec.new_stream_id() for _ in codes]
    parts = [[] for _ in codes]
    t0 = 0
    for t1 in (5, 13, 24):
        for i, c in enumerate(codes):
            parts[i].append(codec.from_indices_tail(c[:, :, :t1].clone(), t0, stream_id=ids[i]))
        t0 = t1
    for i in range(20):
        assert torch.equal(torch.cat(parts[i], -1), full[i]), i
    codec.stream_reset()


# ------------------------------------------------------------------------------- 8f #3 / #4: server batch helpers, engine


def test_server_batch_encode_and_decode_helpers(small_codec):
    """tools/server/model_utils.py:15-86 over MiDAC: batch_encode of ragged wav byte strings == per-item encode;
    batch_vqgan_decode (micro-batches of 8, right padding) == per-item from_indices, trimmed to T_i frames."""
    import io

    import numpy as np
    from scipy.io import wavfile

    from fish_speech_amd import server_utils as SU

    cfg, _, codec = small_codec
    fl = cfg.frame_length
    g = torch.Generator().manual_seed(4)
    wavs = [0.2 * torch.randn(n, generator=g) for n in (3 * fl + 17, fl - 5, 5 * fl)]
    blobs = []
    for w in wavs:
        buf = io.BytesIO()
        wavfile.write(buf, codec.sample_rate, w.numpy().astype(np.float32))
        blobs.append(buf.getvalue())
    feats = SU.batch_encode(codec, blobs)
    assert [f.shape[-1] for f in feats] == [4, 1, 5]
    for w, f in zip(wavs, feats):
        one, n = codec.encode(w.view(1, 1, -1).to(DEV), torch.tensor([w.numel()], device=DEV))
        assert torch.equal(one[0, :, : int(n[0])].cpu(), f)
    assert SU.cached_vqgan_batch_encode(codec, blobs) is SU.cached_vqgan_batch_encode(codec, blobs)
    many = [D.make_codes(cfg, 1, 2 + (i % 5), seed=70 + i)[0] for i in range(11)]      # two micro-batches
    outs = SU.batch_vqgan_decode(codec, many)
    for f, o in zip(many, outs):
        want = codec.from_indices(f[None].clone().to(DEV))[0].cpu().numpy()
        assert o.shape == (1, f.shape[-1] * fl) and float(np.sqrt(np.mean((o - want) ** 2))) <= 1e-5


def test_engine_streaming_segments_equal_final_equal_offline():
    """8f #4 as tested code: the engine-shaped generator (header / segment / final like
    inference_engine/__init__.py:73-140) streams at frame granularity; the concatenated segments are the final audio,
    streaming and non-streaming requests give the same audio, and a one-chunk request equals the offline path
    (generate -> codes[1:, T:-1] -> from_indices under the engine's autocast)."""
    import numpy as np

    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate
    from fish_speech_amd.engine import InferenceResult, StreamingTTSEngine, TTSRequest, inference_wrapper, wav_chunk_header
    from fish_speech_amd.prompt import Conversation, Message, TextPart
    from fish_speech_amd.text2semantic import _system_message
    from oracle import dual_ar as O
    from oracle.fake_tokenizer import ByteTokenizer

    tok = ByteTokenizer()
    cfg = O.DualARConfig(vocab_size=tok.vocab_size + 4, dim=128, n_layer=2, n_head=4, n_local_heads=2, head_dim=32,
                         intermediate_size=256, max_seq_len=2048 + 512, codebook_size=4096, num_codebooks=10,
                         semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                         im_end_id=tok.get_token_id("<|im_end|>"), n_fast_layer=2)
    model = MiDualAR.from_state_dict(DualARConfig.from_any(cfg), O.make_synthetic_state(cfg, seed=5, head_gain=4.0),
                                     device=DEV, im_end_id=cfg.im_end_id)
    model.tokenizer = tok
    model.setup_caches(1, cfg.max_seq_len)
    ccfg = D.DacConfig(encoder_dim=8, decoder_dim=96, n_codebooks=9, codebook_size=1024, semantic_codebook_size=4096,
                       tf_layers=2, tf_window=8, enc_tf_layers=2, enc_tf_window=16)
    codec = MiDAC.from_state_dict(DacConfig.from_any(ccfg), D.make_synthetic_state(ccfg, seed=2), device=DEV)
    engine = StreamingTTSEngine(model, codec)
    text = "<|speaker:0|>First chunk of text.<|speaker:1|>Second chunk, another speaker."
    base = dict(text=text, max_new_tokens=21, chunk_length=30, seed=77, first_chunk_frames=3, chunk_frames=5)

    res = list(engine.inference(TTSRequest(streaming=True, **base)))
    assert [r.code for r in res][0] == "header" and res[-1].code == "final" and all(r.error is None for r in res)
    segs = [r for r in res if r.code == "segment"]
    assert len(segs) >= 4
    hdr = res[0].audio[1].tobytes() if hasattr(res[0].audio[1], "tobytes") else bytes(res[0].audio[1])
    assert wav_chunk_header(sample_rate=codec.sample_rate)[:4] == b"RIFF" and len(wav_chunk_header()) == 44
    assert int.from_bytes(wav_chunk_header(sample_rate=codec.sample_rate)[24:28], "little") == codec.sample_rate
    final = res[-1].audio[1]
    assert np.array_equal(np.concatenate([s.audio[1] for s in segs]), final)
    quiet = list(engine.inference(TTSRequest(streaming=False, **base)))
    assert [r.code for r in quiet] == ["final"] and np.array_equal(quiet[0].audio[1], final)
    out = list(inference_wrapper(TTSRequest(streaming=True, **base), engine))
    assert isinstance(out[0], (bytes, np.ndarray)) and all(isinstance(b, bytes) for b in out[1:-1])
    assert sum(len(b) for b in out[1:-1]) == 2 * final.size          # int16 segments

    # one text chunk == the offline path on the same prompt and seed
    one = "<|speaker:0|>Just one chunk."
    r1 = list(engine.inference(TTSRequest(text=one, max_new_tokens=21, chunk_length=200, seed=5)))
    conv = Conversation([_system_message(None, None), Message(role="user", parts=[TextPart(text=one)]),
                         Message(role="assistant", parts=[], modality="voice", add_im_end=False)])
    prompt, _, _ = conv.encode_for_inference(tok, num_codebooks=10)
    y = generate(model=model, prompt=prompt, max_new_tokens=21, seed=5, temperature=0.8, top_p=0.8, top_k=30)
    codes = y[1:, prompt.shape[1]:-1]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        want = codec.from_indices(codes[None].clone().to(DEV))[0, 0].float().cpu().numpy()
    assert r1[-1].code == "final" and np.array_equal(r1[-1].audio[1], want)
    bad = list(engine.inference(TTSRequest(text=one, max_new_tokens=21, top_p=0.8, prompt_texts=["x"],
                                           prompt_tokens=[torch.zeros(3, 4, dtype=torch.long)])))
    assert bad[-1].code == "error" and isinstance(bad[-1].error, Exception)
    # the engine re-prefills only what a text chunk adds: chunk 2's prompt repeats chunk 1's conversation (prefix-KV reuse)
    model.prefilled_rows = model.reused_rows = 0
    again = list(engine.inference(TTSRequest(streaming=False, **base)))
    assert np.array_equal(again[0].audio[1], final) and model.reused_rows > 0

    # concurrent requests from request threads (tools/api_server.py:115-122) are served one after the other
    # under the model's lock -- every thread gets exactly its sequential result
    import threading

    reqs = [TTSRequest(text=text, max_new_tokens=21, chunk_length=30, seed=70 + i, first_chunk_frames=3, chunk_frames=5,
                       streaming=bool(i % 2)) for i in range(4)]
    want_audio = [[r for r in engine.inference(q)][-1].audio[1] for q in reqs]
    got, errs = [None] * 4, []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            got[i] = [r for r in engine.inference(reqs[i])][-1].audio[1]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(got[i], want_audio[i]), i



@pytest.mark.timeout(240)
def test_batching_engine_concurrent_requests_equal_their_serial_results():
    """engine.BatchingTTSEngine: four request threads at once (two text chunks each, streaming and not) share ONE
    serve_stream loop -- several utterances advance per pass over the weights -- and every thread gets, bit for bit,
    the audio the serial StreamingTTSEngine produces for its request; the slots and the model's lock are free
    afterwards."""
    import threading

    import numpy as np

    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.dual_ar import DualARConfig, MiDualAR
    from fish_speech_amd.engine import BatchingTTSEngine, StreamingTTSEngine, TTSRequest
    from oracle import dual_ar as O
    from oracle.fake_tokenizer import ByteTokenizer

    tok = ByteTokenizer()
    cfg = O.DualARConfig(vocab_size=tok.vocab_size + 4, dim=128, n_layer=2, n_head=4, n_local_heads=2, head_dim=32,
                         intermediate_size=256, max_seq_len=2048 + 512, codebook_size=4096, num_codebooks=10,
                         semantic_begin_id=tok.semantic_begin_id, semantic_end_id=tok.semantic_end_id,
                         im_end_id=tok.get_token_id("<|im_end|>"), n_fast_layer=2)
    model = MiDualAR.from_state_dict(DualARConfig.from_any(cfg), O.make_synthetic_state(cfg, seed=5, head_gain=4.0),
                                     device=DEV, im_end_id=cfg.im_end_id)
    model.tokenizer = tok
    model.setup_caches(4, cfg.max_seq_len)
    ccfg = D.DacConfig(encoder_dim=8, decoder_dim=96, n_codebooks=9, codebook_size=1024, semantic_codebook_size=4096,
                       tf_layers=2, tf_window=8, enc_tf_layers=2, enc_tf_window=16)
    codec = MiDAC.from_state_dict(DacConfig.from_any(ccfg), D.make_synthetic_state(ccfg, seed=2), device=DEV)
    text = "<|speaker:0|>First chunk of text.<|speaker:1|>Second chunk, another speaker."
    reqs = [TTSRequest(text=text, max_new_tokens=21 + 3 * i, chunk_length=30, seed=70 + i, first_chunk_frames=3, chunk_frames=5,
                       streaming=bool(i % 2)) for i in range(4)]
    serial = StreamingTTSEngine(model, codec)
    want = [[r for r in serial.inference(q)] for q in reqs]
    assert all(w[-1].code == "final" for w in want)

    eng = BatchingTTSEngine(model, codec, max_batch=4, step_frames=4)
    in_flight = [0]
    orig_decode = model.decode

    def decode(slots, n):
        in_flight[0] = max(in_flight[0], len(slots))
        return orig_decode(slots, n)

    model.decode = decode
    got, errs = [None] * 4, []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            got[i] = [r for r in eng.inference(reqs[i])]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs and all(g is not None for g in got), errs
    assert in_flight[0] >= 2                  # they really shared passes over the weights
    for q, g, w in zip(reqs, got, want):
        assert g[-1].code == "final" and np.array_equal(g[-1].audio[1], w[-1].audio[1])
        if q.streaming:
            assert g[0].code == "header"
            assert np.array_equal(np.concatenate([r.audio[1] for r in g if r.code == "segment"]), g[-1].audio[1])
    time.sleep(0.2)
    assert model.lock.acquire(blocking=False)
    model.lock.release()
    # the serial engine still works on the same model afterwards (slots were released)
    assert np.array_equal([r for r in serial.inference(reqs[0])][-1].audio[1], want[0][-1].audio[1])
    eng.close()

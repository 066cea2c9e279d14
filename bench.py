#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fish-speech S2 hot path on MI355X.

Metric (BASELINE.json): audio-seconds generated per wall-second, S2-Pro-shaped 4B Dual-AR + DAC codec,
batch 8 utterances per GPU, synthetic 200-token prompts -> 215 frames (10 s of audio) each.
One "step" = one pass of the hot path over one batch: prefill of the 8 prompts, 214 graph-replayed
decode frames (215 frames with the prefill frame), codec decode of the 8x215 frames to waveforms.
Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank r decodes its own batch (utterances r::N of a global batch 8N): no data-path
collective; weights reach ranks > 0 through one RCCL broadcast of the packed arena.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# HBM bytes fetched per decode frame at batch 8 come from a separate rocprofv3 --pmc FETCH_SIZE pass (PMC counters
# cannot be collected inside a timed run): tools/make_profiles.sh runs it and tools/pmc_traffic.py writes the figure,
# with its provenance, into this committed file (KiB x 1024 x 2 = the gfx950 correction of MI355X_MICROARCH.md).
PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def pmc_traffic(batch, frames, int8):
    """-> (bytes per decode frame or None, provenance string)."""
    try:
        with open(PMC_TRAFFIC_JSON) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing"
    if int8 or batch != d.get("batch") or frames != 215:
        return None, "PMC pass was collected for the headline configuration only"
    from fish_speech_amd.build import decode_sources_sha

    have, want = d.get("decode_sources_sha"), decode_sources_sha()
    if have != want:   # the kernels changed after the PMC pass: do not quote its figure as this binary's
        return None, (f"STALE: profiles/pmc_traffic.json ({d['bytes_per_decode_frame']} B per frame) was collected for decode "
                      f"sources {have}, this tree has {want}; re-run tools/make_profiles.sh <tag> pmc")
    return float(d["bytes_per_decode_frame"]), d.get("source", "")

FRAME_LEN = 2048          # samples per frame (modded_dac.py:833,861)
SAMPLE_RATE = 44100
PROMPT_T = 200
N_FRAMES = 215            # 10 s of audio
BATCH = 8


def s2_pro_config(max_seq_len=1024):
    from fish_speech_amd.dual_ar import DualARConfig

    # ASSUMPTION: S2-Pro's config.json is not in the reference repo; widths follow the
    # README ("4B slow / 400M fast / 10 codebooks") and Qwen3-4B.
    return DualARConfig(vocab_size=155776, n_layer=36, n_head=32, n_local_heads=8, head_dim=128, dim=2560,
                        intermediate_size=9728, codebook_size=4096, num_codebooks=10, semantic_begin_id=151678,
                        semantic_end_id=151678 + 4095, im_end_id=151645, max_seq_len=max_seq_len,
                        rope_base=1000000.0, norm_eps=1e-6, attention_qk_norm=True, scale_codebook_embeddings=True,
                        norm_fastlayer_input=True, n_fast_layer=4)


def synthetic_state_on_device(cfg, device, seed=0):
    """Random-init weights of the S2-Pro shape, generated on the GPU (N(0, 0.02), norms ~1)."""
    from fish_speech_amd.dual_ar import expected_state_shapes

    g = torch.Generator(device=device).manual_seed(seed)
    st = {}
    for name, shape in expected_state_shapes(cfg).items():
        if len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            t = 0.02 * torch.randn(shape, generator=g, device=device)
        st[name] = t.to(torch.bfloat16)
    return st


def quantize_state_int8_on_device(state):
    """WeightOnlyInt8QuantHandler.create_quantized_state_dict (tools/llama/quantize.py:186-202) on the GPU tensors:
    every 2-D non-embedding weight -> int8 + per-row bf16 scales (max|row| / 127.5)."""
    out = {}
    for k, v in state.items():
        if v.dim() == 2 and k.endswith(".weight") and "embeddings" not in k:
            x = v.float()
            amax = x.abs().amax(dim=1)
            scales = torch.clamp(amax / 127.5, min=torch.finfo(torch.float32).eps)
            out[k] = torch.clamp(torch.round(x / scales[:, None]), -128, 127).to(torch.int8)
            out[k[: -len("weight")] + "scales"] = scales.to(torch.bfloat16)
        else:
            out[k] = v
    return out


def make_prompts(cfg, n, base_seed):
    ps = []
    for i in range(n):
        g = torch.Generator().manual_seed(base_seed + i)
        p = torch.zeros(cfg.num_codebooks + 1, PROMPT_T, dtype=torch.int64)
        p[0] = torch.randint(0, 150000, (PROMPT_T,), generator=g)
        ps.append(p)
    return ps


def algorithmic_bytes_per_frame(cfg, batch, mean_ctx, int8=False):
    """weight bytes streamed once per frame step (bf16; int8 + a bf16 scale per row for --int8) +
    per-utterance KV reads (bf16)."""
    d, ffn = cfg.dim, cfg.intermediate_size
    qkv = (cfg.n_head + 2 * cfg.n_local_heads) * cfg.head_dim
    per_layer = qkv * d + d * cfg.n_head * cfg.head_dim + 3 * ffn * d
    slow = cfg.n_layer * per_layer
    fast = cfg.num_codebooks * cfg.n_fast_layer * per_layer
    # fast layer 0 sees fast_embeddings[code] at codebook positions 1..ncb-1: its wqkv output is a pure function of the
    # code and is read from a per-code table (one row per utterance) instead of streaming the matrix
    fast -= (cfg.num_codebooks - 1) * qkv * d
    n_live = cfg.semantic_end_id - cfg.semantic_begin_id + 2
    heads = n_live * d + (cfg.num_codebooks - 1) * cfg.codebook_size * d
    kv = batch * cfg.n_layer * 2 * cfg.n_local_heads * cfg.head_dim * mean_ctx
    if int8:   # one byte per weight, two per output row (scale); rows ~ weights / d
        w = slow + fast + heads
        return w + 2 * (w // d) + 2 * kv
    return 2 * (slow + fast + heads + kv)


def synthetic_codec_state(cfg, device, seed=1):
    """Random-init codec weights of the yaml's shape, generated on the GPU (no checkpoint is available)."""
    from fish_speech_amd.dac import expected_state_shapes as codec_shapes

    g = torch.Generator(device=device).manual_seed(seed)
    st = {}
    for name, shape in codec_shapes(cfg).items():
        if name.endswith("alpha"):
            t = 0.5 + torch.rand(shape, generator=g, device=device)
        elif name.endswith(".gamma"):
            t = 0.3 + 0.05 * torch.randn(shape, generator=g, device=device)
        elif name.endswith("norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g, device=device)
        elif name.endswith("codebook.weight"):
            t = torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g, device=device) * (0.75 / max(fan_in, 1) ** 0.5)
        st[name] = t
    return st


def construct(cfg, device, rank, int8=False, with_codec=True, replicate=None):
    """The objects of one rank, in the order every rank of an N-GPU job builds them: handle over an empty arena; rank 0
    alone generates / loads the weights (a checkpoint is read once per node); `replicate(obj)` (dist.broadcast_arena)
    puts the arena content in place on the other ranks and marks it ready there -- those ranks never see load_tensor or
    finalize, their derived tables (fast layer-0 q|k|v, row-balanced copies) are rebuilt from the received bytes by the
    first prefill.  Returns (model, codec, state, codec_state); the states are None on ranks > 0.
    tests/test_rank_gpu.py runs this with rank = 1 inside the 1-GPU lease."""
    from fish_speech_amd.dual_ar import MiDualAR

    model = MiDualAR(cfg, device=device, im_end_id=cfg.im_end_id)
    state = None
    if rank == 0:
        state = synthetic_state_on_device(cfg, device)
        model.load_state_dict(quantize_state_int8_on_device(state) if int8 else state)
    if replicate is not None:
        replicate(model)   # (dist.broadcast_arena orders the model's stream after the collective: no host sync here)
    # 1024 positions per slot: covers config 3's 400 + 430.  Sixteen slots (the handle's caches are set up once): the timed
    # region runs 8 utterances in 8 of them, other_configs.batch16 all sixteen
    model.setup_caches(2 * BATCH, cfg.max_seq_len)
    model.set_ignore_eos(True)
    codec, codec_state = None, None
    if with_codec:
        from fish_speech_amd.dac import DacConfig, MiDAC

        ccfg = DacConfig()
        codec = MiDAC(ccfg, device=device)
        if rank == 0:
            codec_state = synthetic_codec_state(ccfg, device)
            codec.load_folded_state(codec_state)
        if replicate is not None:
            replicate(codec)
    return model, codec, state, codec_state


def run_step(model, codec, prompts, samp_seeds, device):
    """One pass of the hot path over one batch: prefill + 214 graph-replayed decode frames + codec decode
    of the generated codes (rows 1.. of each frame; the last frame is dropped like generate_long does,
    inference.py:708, and replaced by ... nothing: 215 frames are generated and 215 are decoded here)."""
    from fish_speech_amd.dual_ar import generate_batch_device

    codes = generate_batch_device(model=model, prompts=prompts, max_new_tokens=N_FRAMES, seeds=samp_seeds,
                                  temperature=0.7, top_p=0.7, top_k=30)   # (B, ncb, N_FRAMES) int64 on device
    wav = codec.from_indices(codes) if codec is not None else None
    return codes, wav


def run_steps_overlapped(model, codec, prompts, samp_seeds, device, steps, side):
    """`steps` passes of the hot path with the two halves of a step overlapped ACROSS steps: while the HBM-bound frame
    loop of batch i replays on the model's stream, the MFMA-bound codec decode of batch i - 1 runs on the codec's own
    stream (`codec.set_async(True)`: no cross-queue wait is left pending).  Every step's prefill, 214 frames and codec
    decode happen inside the call; the last batch's decode is not overlapped with anything.  Returns (per-step decode
    frame ms, the last batch's waveform).  `side`: a torch stream the codec's buffers are allocated on."""
    from fish_speech_amd.dual_ar import finish_batch_device, generate_batch_device

    frame_ms, pending, wav, keep = [], None, None, []
    for _ in range(steps):
        tok = generate_batch_device(model=model, prompts=prompts, max_new_tokens=N_FRAMES, seeds=samp_seeds,
                                    temperature=0.7, top_p=0.7, top_k=30, wait=False)    # enqueued, host free
        if pending is not None:
            codec.synchronize()                                     # (the previous decode's buffers may be recycled now)
            side.wait_stream(torch.cuda.current_stream(device))     # the codes were copied out on the current stream
            with torch.cuda.stream(side):
                wav = codec.from_indices(pending)
            keep = [pending, wav]                                   # alive until the codec stream is done with them
        pending = finish_batch_device(model, tok)                   # host waits for the frames here
        ms, _ = model.last_decode_stats()
        frame_ms.append(ms / max(N_FRAMES - 1, 1))
    codec.synchronize()
    wav = codec.from_indices(pending)
    codec.synchronize()
    del keep
    return frame_ms, wav


def cpu_baseline(cfg, state_dev, codec_state_dev, n_frames=64, codec_frames=N_FRAMES):
    """The oracle (CPU restatement of the reference path, kind 'port') timed on this box's host cores on a
    bounded sample of the same workload: Dual-AR prefill of one 200-token prompt + n_frames decode frames at context
    200.. (batch 1 -- the reference cannot batch), timed in two halves so that the growth of a frame with its context
    is measured, not assumed; extrapolated along that trend to 215 frames (the flat extrapolation is reported beside it),
    and the codec decode of the FULL 215 frames of one utterance."""
    from oracle import dac as OD
    from oracle import dual_ar as O

    oc = O.s2_pro_shaped_config(max_seq_len=512)
    oc.semantic_begin_id, oc.semantic_end_id, oc.im_end_id = cfg.semantic_begin_id, cfg.semantic_end_id, cfg.im_end_id
    st = {k: v.cpu() for k, v in state_dev.items()}
    orc = O.DualAROracle(oc, st)
    prompt = make_prompts(cfg, 1, 1000)[0]
    orc.setup_caches(1, oc.max_seq_len)
    # the bf16 CPU path is memory-bound and oversubscribes badly: sweep the thread count on two decode frames each
    # and time the sample with the best one (the reference's own default is torch's, i.e. every core)
    all_threads = torch.get_num_threads()
    sweep = {}
    O.generate(orc, prompt[:, :8], 1, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(1, 0), stop_on_im_end=False)  # page in
    for nt in sorted({t for t in (8, 16, 32, 64, all_threads) if t <= all_threads}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        O.generate(orc, prompt[:, :8], 3, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(1, 0), stop_on_im_end=False)
        sweep[nt] = (time.perf_counter() - t0) / 3
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    O.generate(orc, prompt, 1, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(1, 0), stop_on_im_end=False)
    t_prefill = time.perf_counter() - t0
    half = n_frames // 2
    t0 = time.perf_counter()
    O.generate(orc, prompt, 1 + half, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(1, 0), stop_on_im_end=False)
    t_half = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.generate(orc, prompt, 1 + n_frames, 0.7, 0.7, 30, uniform_fn=O.FmiUniform(1, 0), stop_on_im_end=False)
    t_all = time.perf_counter() - t0
    per_frame = max(t_all - t_prefill, 1e-9) / n_frames
    pf_a = max(t_half - t_prefill, 1e-9) / half                  # frames [0, half): mean context PROMPT_T + half / 2
    pf_b = max(t_all - t_half, 1e-9) / (n_frames - half)         # frames [half, n): mean context PROMPT_T + 3 half / 2
    # seconds per frame per frame of context.  A frame cannot get cheaper with a longer context: a negative measured slope
    # is timer / scheduler noise between the two halves (seen: 0.262 then 0.227 s/frame) and is read as zero, i.e. flat
    slope = max((pf_b - pf_a) / half, 0.0)
    t_flat = t_prefill + (N_FRAMES - 1) * per_frame
    t_ar = t_prefill + sum(max(pf_a + slope * (i - half / 2), 0.0) for i in range(N_FRAMES - 1))
    del orc, st
    t_codec, note, cbest = 0.0, "codec not included", best
    if codec_state_dev is not None:
        ccfg = OD.DacConfig()
        cst = {k: v.float().cpu() for k, v in codec_state_dev.items()}
        corc = OD.DacOracle(ccfg, cst)
        with torch.no_grad():
            csweep = {}
            small = OD.make_codes(ccfg, 1, 16, seed=1)
            for nt in sorted({t for t in (16, 32, 64, all_threads) if t <= all_threads}):
                torch.set_num_threads(nt)
                t0 = time.perf_counter()
                corc.from_indices(small.clone())
                csweep[nt] = time.perf_counter() - t0
            cbest = min(csweep, key=csweep.get)
            torch.set_num_threads(cbest)
            codes = OD.make_codes(ccfg, 1, codec_frames, seed=1)
            t0 = time.perf_counter()
            corc.from_indices(codes.clone())
            t_c = time.perf_counter() - t0
        t_codec = t_c * N_FRAMES / codec_frames
        note = f"codec decode of {codec_frames} frames {t_c:.2f}s (fp32, {cbest} threads)"
    torch.set_num_threads(all_threads)
    return {
        "value": round(10.0 / (t_ar + t_codec), 5), "unit": "audio-sec/s", "cores": max(best, cbest),
        "cores_dual_ar": best, "cores_codec": cbest,   # each leg runs with the best thread count of its own sweep
        "kind": "port",
        "sample": f"oracle = the bit-equal CPU port of the reference path (tests/test_oracle_cpu.py; /root/reference itself "
                  f"is not on this box), torch CPU, batch 1, Dual-AR on {best} threads (best of sweep "
                  f"{ {k: round(v, 2) for k, v in sweep.items()} } s/frame-ish, host has {os.cpu_count()} cpus): "
                  f"prefill {PROMPT_T} tokens {t_prefill:.2f}s + {n_frames} decode "
                  f"frames at context {PROMPT_T}..{PROMPT_T + n_frames} at {per_frame:.3f}s/frame (bf16; first half "
                  f"{pf_a:.3f}, second half {pf_b:.3f}), extrapolated along that trend to {N_FRAMES} frames = {t_ar:.1f}s "
                  f"(flat extrapolation {t_flat:.1f}s, {100 * (t_flat / t_ar - 1):+.1f} %); {note}",
    }


def measure_prefill(model, prompts, seeds, reps=3):
    """Median wall time (HIP events on torch's stream, which the library orders itself against) of the prefill call of
    the step: 8 x 200 prompt rows through 36 layers (tiled MFMA GEMM + MFMA flash attention) plus the first frame's
    head, sampler and fast-AR chain."""
    import statistics

    sp = [model._sampling(0.7, 0.7, 30, s, True) for s in seeds]
    slots = list(range(len(prompts)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0.record()
        model.prefill(slots, prompts, [2] * len(prompts), sp)
        model.wait_stream()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
        for i in slots:
            model.release(i)
    return statistics.median(ts[1:])


def prefill_roofline(cfg, prefill_ms):
    """MFMA-bound part of the step: the slow transformer over the prompt rows.  flops = 2 x slow-layer weights x rows
    + causal attention 4 H D T^2 / 2 per prompt and layer; the measured time also contains the first
    frame's fast-AR chain (HBM-bound, ~half a decode frame), so `frac` is a lower bound for the GEMM + attention."""
    rows = BATCH * PROMPT_T
    d, ffn = cfg.dim, cfg.intermediate_size
    qkv = (cfg.n_head + 2 * cfg.n_local_heads) * cfg.head_dim
    per_layer = qkv * d + d * cfg.n_head * cfg.head_dim + 3 * ffn * d
    flops = 2.0 * cfg.n_layer * per_layer * rows + BATCH * cfg.n_layer * 4.0 * cfg.n_head * cfg.head_dim * PROMPT_T * PROMPT_T / 2
    ach = flops / (prefill_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
            "kernel": "prefill: linear_tiled_256p_kernel<0|1|2, MF> + attn_prefill_mfma_kernel over 8 x 200 rows, 36 layers "
                      "(time includes the first frame's head + fast-AR chain)",
            "flops_per_launch": flops, "avg_launch_ms": round(prefill_ms, 3), "traffic": None}


def codec_roofline(codec_ms):
    """Codec decode of 8 x 215 frames on the fp16 matrix cores: useful flops (1.454 TFLOP per 215-frame
    utterance) and issued flops (the two-term fp16 split runs 3 MFMA products per useful one)."""
    useful = BATCH * 1.454e12 * N_FRAMES / 215
    ach = useful / (codec_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 1), "issued": round(3 * ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(ach / 2500.0, 4), "frac_issued": round(3 * ach / 2500.0, 4),
            "kernel": "conv_mfma_bf16_kernel<..., NP = 2> (fp16 two-term split, fp32-class results) + linear_planes_kernel",
            "flops_per_launch": useful, "avg_launch_ms": round(codec_ms, 3), "traffic": None}


def measure_encode(codec, device):
    """DAC.encode throughput (extract_vq.py's job): one 3 s clip and a batch of 64 x 3 s clips (fp32 matrix cores)."""
    g = torch.Generator(device=device).manual_seed(7)
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, b in (("clip_3s", 1), ("batch64_3s", 64)):
        a = 0.1 * torch.randn(b, 1, 3 * SAMPLE_RATE, generator=g, device=device)
        lens = torch.full((b,), 3 * SAMPLE_RATE, device=device, dtype=torch.long)
        codec.encode(a, lens)
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            e0.record()
            codec.encode(a, lens)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        out[name] = {"ms": round(ms, 2), "audio_sec_per_s": round(3.0 * b / (ms * 1e-3), 1),
                     "tflops_useful": round(2 * 115.8e9 * b / (ms * 1e-3) / 1e12, 1)}
    out["note"] = "fp32 matrix cores (v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s); 115.8 useful GMAC per 3 s clip (SURVEY 8d)"
    return out


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


def mixed_length_workload(cfg, n_utts, seed=2024):
    """BASELINE.json config 4 / SURVEY.md 8d: prompts T ~ U{50..400}, frames ~ U{100..430} per utterance."""
    g = torch.Generator().manual_seed(seed)
    prompts, frames = [], []
    for _ in range(n_utts):
        T = int(torch.randint(50, 401, (1,), generator=g))
        n = int(torch.randint(100, 431, (1,), generator=g))
        p = torch.zeros(cfg.num_codebooks + 1, T, dtype=torch.int64)
        p[0] = torch.randint(0, 150000, (T,), generator=g)
        prompts.append(p)
        frames.append(n)
    return prompts, frames


def run_config4(model, codec, cfg, rank, world, dist, device, per_gpu=BATCH):
    """Config 4: a global queue of per_gpu x world mixed-length utterances (BASELINE: 8 per GPU = batch 64 over 8 GPUs),
    LPT-packed over the ranks by expected frames (scheduler.partition_for_ranks -- no data-path collective), each rank
    running its share through continuous batching (8 slots, finished slots refilled) and decoding its utterances' codes
    with the codec in ragged batches (MiDAC.from_indices_ragged).  per_gpu = 16: two utterances per slot, so the slots
    refill (with 8 per GPU the loop runs the longest utterance's frames at a falling occupancy whatever the scheduler)."""
    from fish_speech_amd.scheduler import generate_queue, lpt_order, partition_for_ranks

    prompts, frames = mixed_length_workload(cfg, per_gpu * world)
    mine = partition_for_ranks([f + 0.1 * p.shape[1] for p, f in zip(prompts, frames)], world)[rank]
    ps, fr = [prompts[i] for i in mine], [frames[i] for i in mine]
    seeds = [9000 + i for i in mine]

    def once():
        res = generate_queue(model=model, prompts=ps, max_new_tokens=fr, max_batch=BATCH, seeds=seeds,
                             order=lpt_order(fr), temperature=0.7, top_p=0.7, top_k=30)
        n = 0
        todo = []
        for r, p, f in zip(res, ps, fr):
            codes = r[1:, p.shape[1]:].to(device).contiguous()
            assert codes.shape[-1] == f
            todo.append(codes)
            n += f
        if codec is not None:   # ragged batched decode: groups of up to 8 utterances of similar length per call
            codec.from_indices_ragged(todo)
        _sync(device)
        return n

    once()                                   # warm-up (graphs for every live-slot count)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    n_frames = once()
    dt = time.perf_counter() - t0
    # (reductions travel as device tensors over RCCL; in the one-GPU gloo debug mode as HOST tensors: gloo's device-tensor
    # path between ranks that share a GPU is what profiles/r06_startup_order_stress.txt found faulting)
    t = torch.tensor([dt, dt, float(n_frames)], device="cpu" if ONE_GPU_DEBUG else device, dtype=torch.float64)
    if dist:
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt_max, dt_mean, total = float(mx[0]), float(t[1]) / world, float(t[2])
    else:
        dt_max, dt_mean, total = dt, dt, float(n_frames)
    return {"workload": f"configs[3]: {per_gpu * world} mixed-length utterances (T~U[50,400], frames~U[100,430]) over "
                        f"{world} GPU(s), LPT-partitioned, continuous batching with {BATCH} slots + ragged batched codec decode",
            "audio_sec_per_s": round(total * FRAME_LEN / SAMPLE_RATE / dt_max, 2), "wall_s": round(dt_max, 3),
            "frames_total": int(total),
            "rank_imbalance_max_over_mean": round(dt_max / dt_mean, 4)}


def run_batch16(model, codec, cfg, device):
    """The bench step with 16 utterances per GPU instead of 8 (NOT the headline, BASELINE.json
    quotes batch 8): the decode GEMVs take their rows in two sets of 8 (linear_skinny_kernel<..., XR = 16>), one pass
    over the weights serves all 16.  Same prompts shape, same sampler, codec decode of all 16 inside the timed step."""
    prompts = make_prompts(cfg, 16, 5000)
    seeds = [9000 + i for i in range(16)]
    run_step(model, codec, prompts, seeds, device)
    _sync(device)
    t0 = time.perf_counter()
    codes, _ = run_step(model, None, prompts, seeds, device)
    ms, launches = model.last_decode_stats()
    codec.from_indices(codes)
    _sync(device)
    dt = time.perf_counter() - t0
    return {"batch_per_gpu": 16, "audio_sec_per_s": round(16 * N_FRAMES * FRAME_LEN / SAMPLE_RATE / dt, 2),
            "ms_per_step": round(dt * 1e3, 1), "decode_frame_avg_ms": round(ms / (N_FRAMES - 1), 4),
            "launches_per_frame": launches, "note": "codec decode of the 16 utterances inside the step; not the headline"}


def run_overlapped(model, codec, prompts, seeds, device, steps=3):
    """VERDICT r04 #2, reported beside the serial step so rounds stay comparable: the same step with the codec decode of
    batch i - 1 running BESIDE the frame loop of batch i (run_steps_overlapped).  Measured after the timed region; NOT what
    `value` is: on this hardware the two queues time-slice the CUs (a conv work-group holds half a CU's registers for
    milliseconds, the loop's 6-20 us GEMVs queue behind it) and the frames lose what the codec gains
    (profiles/r05_overlap_step.txt, r05_overlap_floor.txt)."""
    side = torch.cuda.Stream(device=device)
    codec.set_async(True)
    try:
        run_steps_overlapped(model, codec, prompts, seeds, device, 1, side)
        _sync(device)
        t0 = time.perf_counter()
        fms, _ = run_steps_overlapped(model, codec, prompts, seeds, device, steps, side)
        _sync(device)
        dt = (time.perf_counter() - t0) / steps
    finally:
        codec.set_async(False)
    return {"ms_per_step": round(dt * 1e3, 2), "audio_sec_per_s": round(BATCH * N_FRAMES * FRAME_LEN / SAMPLE_RATE / dt, 2),
            "decode_frame_avg_ms": round(sum(fms) / len(fms), 4), "steps": steps,
            "note": "codec decode of batch i-1 on its own stream beside the frame loop of batch i; measured and not adopted"}


def run_config1(model, codec, cfg, device):
    """Config 1 (BASELINE configs[1]): one utterance, greedy, 200-token prompt -> 215 frames + codec decode."""
    from fish_speech_amd.dual_ar import generate_batch_device

    p = make_prompts(cfg, 1, 5000)

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        codes = generate_batch_device(model=model, prompts=p, max_new_tokens=N_FRAMES, seeds=[1], temperature=0.7,
                                      top_p=0.7, top_k=1)
        ms, _ = model.last_decode_stats()
        if codec is not None:
            codec.from_indices(codes)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ms / max(N_FRAMES - 1, 1)

    once()
    dt, frame_ms = min(once() for _ in range(2))
    return {"workload": "configs[1]: single utterance greedy decode + codec, batch=1", "latency_s": round(dt, 4),
            "audio_sec_per_s": round(N_FRAMES * FRAME_LEN / SAMPLE_RATE / dt, 2), "decode_frame_ms": round(frame_ms, 4)}


def run_config5(model, codec, cfg, device, runs=5, first=8, chunk=32):
    """Config 5: streaming chunked decode at batch 8 -- p50 wall time until the first audio chunk is complete."""
    import statistics

    from fish_speech_amd.stream import generate_stream

    prompts = make_prompts(cfg, BATCH, 1000)
    seeds = [4242 + i for i in range(BATCH)]

    def once(growth=1.0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first_t, n = None, 0
        for ch in generate_stream(model=model, codec=codec, prompts=prompts, max_new_tokens=N_FRAMES + 1,
                                  first_chunk_frames=first, chunk_frames=chunk, seeds=seeds, temperature=0.7,
                                  top_p=0.7, top_k=30, chunk_growth=growth):
            torch.cuda.synchronize()
            if first_t is None:
                first_t = time.perf_counter() - t0
            n += ch.audio.shape[-1]
        return first_t, time.perf_counter() - t0, n

    once()
    rs = [once() for _ in range(runs)]
    tot = statistics.median(r[1] for r in rs)
    once(2.0)
    grow = statistics.median(once(2.0)[1] for _ in range(3))
    return {"workload": f"configs[4]: streaming, batch=8, first chunk {first} frames then every {chunk}",
            "first_audio_ms_p50": round(statistics.median(r[0] for r in rs) * 1e3, 2),
            "stream_audio_sec_per_s": round(BATCH * rs[0][2] / SAMPLE_RATE / tot, 2),
            "stream_audio_sec_per_s_growing_chunks": round(BATCH * rs[0][2] / SAMPLE_RATE / grow, 2)}


def run_config5_staggered(model, codec, cfg, n_req=16, gap_s=0.25):
    """Configs 4 + 5 together (serving.serve_stream): requests ARRIVE over time (one every gap_s), 8 slots, every
    utterance streamed on its own chunk schedule, slots refilled as utterances finish.  Reports first-audio latency
    from arrival (p50 / p90 over the requests, queueing for a slot included) and whole-run throughput.  gap_s = 0.25:
    40 audio-seconds demanded per second, about 0.8 of what this loop sustains (at 0.12 s the queue grows without
    bound and the latency is queueing time: p50 194 ms, p90 399 ms measured).  Since round 4 the requests are put by a
    producer thread into a RequestFeed (no knowledge of future arrivals: while a slot is free the live utterances advance
    `open_slot_step` frames at a time); rounds 2-3 passed a list with known arrival times (55 / 62 ms p50 / p90)."""
    import statistics
    import threading

    from fish_speech_amd.serving import RequestFeed, StreamRequest, serve_stream

    prompts = make_prompts(cfg, n_req, 7000)

    def once():
        # The requests come from a producer THREAD through a RequestFeed, as a server's request threads would put them:
        # the loop knows nothing about future arrivals (the round-3 figure was measured over a request list
        # with known arrival times, whose `admit_early` cut is clairvoyant).  Latency counts from the put.
        reqs = [StreamRequest(prompt=p, max_new_tokens=N_FRAMES + 1, seed=8000 + i, rid=i) for i, p in enumerate(prompts)]
        feed = RequestFeed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()

        def producer():
            for i, r in enumerate(reqs):
                dt = t0 + i * gap_s - time.perf_counter()
                if dt > 0:
                    time.sleep(dt)
                feed.put(r)
            feed.close()

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        first, samples = {}, 0
        for ev in serve_stream(model=model, codec=codec, requests=feed, max_batch=BATCH, step_frames=8,
                               first_chunk_frames=8, chunk_frames=32, chunk_growth=2.0, temperature=0.7, top_p=0.7, top_k=30):
            if ev.kind == "segment":
                torch.cuda.synchronize()
                samples += ev.audio.shape[-1]
                if ev.first_audio_latency is not None:
                    first[ev.rid] = time.perf_counter() - reqs[ev.rid].arrival_abs
        th.join()
        return sorted(first.values()), samples, time.perf_counter() - t0

    once()
    lat, samples, wall = once()
    return {"workload": f"configs[3]+[4]: {n_req} requests arriving every {int(gap_s * 1e3)} ms, {BATCH} slots, per-utterance "
                        f"streaming (first chunk 8 frames, then 32, 64, ...) with slot refill (serving.serve_stream over a "
                        f"RequestFeed filled by a producer thread: arrivals unknown to the loop)",
            "first_audio_ms_p50": round(statistics.median(lat) * 1e3, 1),
            "first_audio_ms_p90": round(lat[int(0.9 * (len(lat) - 1))] * 1e3, 1),
            "audio_sec_per_s": round(samples / SAMPLE_RATE / wall, 2), "wall_s": round(wall, 3)}


# FMI_BENCH_ONE_GPU=1: a DEBUG mode for boxes with one GPU -- every rank of `--gpus N` uses cuda:0 and the process group is
# gloo (RCCL refuses two ranks on one device; gloo is handed host tensors, dist.broadcast_buffer).  It executes bench.py's own N > 1 path end to
# end -- the re-exec under torch.distributed.run, per-rank construction, both arena broadcasts, the barriers, the MAX / SUM
# reductions, rank 0's JSON line -- with the ranks time-slicing one GPU, so its numbers are NOT a result (the line says so).
ONE_GPU_DEBUG = os.environ.get("FMI_BENCH_ONE_GPU", "0") not in ("", "0")


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n and not ONE_GPU_DEBUG:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node -- refusing to report a "
                         f"{n}-GPU number from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if ONE_GPU_DEBUG:
        env.setdefault("GLOO_SOCKET_IFNAME", "lo")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--int8", action="store_true",
                    help="weight-only int8 checkpoint of the same model (NOT the headline number: reduced precision)")
    ap.add_argument("--no-codec", action="store_true", help="debug: Dual-AR only (INVALID as a result)")
    ap.add_argument("--frames", type=int, default=215, help="debug: fewer frames (INVALID as a result)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (backend nccl = RCCL) even at --gpus 1 and run the arena broadcast "
                         "and the reductions of the N > 1 path through it (exercises RCCL inside a 1-GPU lease)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra measurements (configs 1, 3, 4 of BASELINE.json) after the timed region")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda:0" if ONE_GPU_DEBUG else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:   # --force-dist without a launcher: a one-rank rendezvous on loopback
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        if ONE_GPU_DEBUG:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    # the library ships prebuilt in-tree; if it is stale only one process per node compiles it
    from fish_speech_amd.build import build
    if local_rank == 0:
        build(verbose=(rank == 0))
    if dist:
        dist.barrier()
    from fish_speech_amd.dual_ar import MiDualAR

    globals()["N_FRAMES"] = args.frames
    cfg = s2_pro_config()
    cfg.weight_int8 = bool(args.int8)
    replicate = None
    if dist:  # the only collective of the path: one broadcast of each packed weight arena (xGMI)
        from fish_speech_amd.dist import broadcast_arena

        replicate = lambda m: broadcast_arena(m, src=0)   # noqa: E731
    model, codec, state, codec_state = construct(cfg, device, rank, int8=args.int8, with_codec=not args.no_codec,
                                                 replicate=replicate)
    if ONE_GPU_DEBUG and dist:   # debug mode only: did every rank receive rank 0's arenas?  (sum of the bytes as int64 words)
        sums = [int(o.arena[: o.arena.numel() // 8 * 8].view(torch.int64).sum().item()) for o in (model, codec) if o is not None]
        gathered = [None] * world
        dist.all_gather_object(gathered, sums)
        if any(g != gathered[0] for g in gathered):
            raise SystemExit(f"rank {rank}: arena checksums differ across ranks after the broadcast: {gathered}")
        if rank == 0:
            print(f"[one-GPU debug] arena checksums equal on all {world} ranks: {gathered[0]}", file=sys.stderr, flush=True)
    if args.no_graph:
        model.set_graph(False)
    prompts = make_prompts(cfg, BATCH, 1000 + rank * BATCH)
    seeds = [4242 + rank * BATCH + i for i in range(BATCH)]

    def stage(what):   # (debug mode only: where a rank was, should it die)
        if ONE_GPU_DEBUG:
            print(f"[one-GPU debug] rank {rank}: {what}", file=sys.stderr, flush=True)

    stage("constructed")
    for _ in range(args.warmup):
        run_step(model, codec, prompts, seeds, device)
    torch.cuda.synchronize()
    stage("warm-up done")
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frame_ms = []
    codec_ms = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.steps):
        codes, _ = run_step(model, None, prompts, seeds, device)
        ms, launches = model.last_decode_stats()
        frame_ms.append(ms / max(N_FRAMES - 1, 1))
        if codec is not None:
            ev0.record()
            codec.from_indices(codes)
            ev1.record()
            ev1.synchronize()
            codec_ms.append(ev0.elapsed_time(ev1))
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cpu" if ONE_GPU_DEBUG else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # untimed side measurements for the other two rooflines (after the timed region, same objects)
    prefill_ms = measure_prefill(model, prompts, seeds) if N_FRAMES == 215 else None   # (not in --frames debug / PMC passes)
    audio_s = world * BATCH * N_FRAMES * FRAME_LEN / SAMPLE_RATE * args.steps
    mean_ctx = PROMPT_T + N_FRAMES / 2
    bytes_frame = algorithmic_bytes_per_frame(cfg, BATCH, mean_ctx, int8=args.int8)
    avg_frame_s = (sum(frame_ms) / len(frame_ms)) * 1e-3
    achieved = bytes_frame / avg_frame_s / 1e9
    out = {
        "metric": "audio-sec/s (inverse RTF), S2-Pro-shaped 4B Dual-AR batch=8 per GPU, 200-token prompts -> 10 s audio",
        "value": round(audio_s / dt, 3),
        "unit": "audio-sec/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16 activations, int8 weights (NOT the headline configuration)" if args.int8 else "bf16",
        "data": ("INVALID AS A RESULT (FMI_BENCH_ONE_GPU debug mode: every rank on cuda:0 over gloo) -- " if ONE_GPU_DEBUG else "")
                + "synthetic (random-init S2-Pro-shaped weights, random 200-token prompts, EOS ignored, 215 frames)",
        "config": {"workload": "configs[2]: S2-Pro 4B batch=8, 200-token prompts -> 10 s audio, hipGraph inner-AR loop",
                   "batch_per_gpu": BATCH, "prompt_tokens": PROMPT_T, "frames": N_FRAMES, "parallelism": f"utterance-sharded x{world}",
                   "codec_in_step": codec is not None},
        "semantic_frames_per_s": round(world * BATCH * N_FRAMES * args.steps / dt, 1),
        "breakdown_ms": {"decode_frame_avg": round(avg_frame_s * 1e3, 4), "launches_per_frame": launches,
                         "codec_decode_batch": round(sum(codec_ms) / len(codec_ms), 2) if codec_ms else None},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(achieved / 8000.0, 4),
                     "traffic": pmc_traffic(BATCH, N_FRAMES, args.int8)[0],
                     "traffic_source": pmc_traffic(BATCH, N_FRAMES, args.int8)[1],
                     # what the binary really moved per frame (PMC) / the same time / the same peak: fast positions 0 and 1
                     # share a pass over the fast weights since round 4, so it streams LESS than the SURVEY 8d formula
                     # `achieved` divides by the same time -- `frac` credits bytes that are no longer read
                     "traffic_frac": (round(pmc_traffic(BATCH, N_FRAMES, args.int8)[0] / avg_frame_s / 1e9 / 8000.0, 4)
                                      if pmc_traffic(BATCH, N_FRAMES, args.int8)[0] else None),
                     "kernel": f"decode frame: {launches} launches replayed as one hipGraph -- linear_skinny_kernel (weight "
                               "streaming: 36 slow layers x 4 + the fast transformer's 9 passes x 4 layers x 4 + heads), 36 "
                               "attention, 36 fast-attention, 10 sampler launches",
                     "bytes_per_launch": bytes_frame, "avg_launch_ms": round(avg_frame_s * 1e3, 4)},
    }
    if prefill_ms is not None:
        out["roofline_prefill"] = prefill_roofline(cfg, prefill_ms)
    if codec_ms:
        out["roofline_codec"] = codec_roofline(sum(codec_ms) / len(codec_ms))
    if not args.no_extras and N_FRAMES == 215 and codec is not None and world == 1:
        out["encode"] = measure_encode(codec, device)
    if not args.no_extras and N_FRAMES == 215:
        # untimed extras: the other single-node configurations of BASELINE.json, measured with the same objects
        extras = {"config3_mixed_lengths": run_config4(model, codec, cfg, rank, world, dist, device),
                  "config3_mixed_lengths_queue16": run_config4(model, codec, cfg, rank, world, dist, device, per_gpu=2 * BATCH)}
        if world == 1 and codec is not None:
            extras["config1_batch1_greedy"] = run_config1(model, codec, cfg, device)
            extras["config4_streaming"] = run_config5(model, codec, cfg, device)
            extras["config4_streaming_staggered_arrivals"] = run_config5_staggered(model, codec, cfg)
            extras["batch16"] = run_batch16(model, codec, cfg, device)
            extras["overlapped_step"] = run_overlapped(model, codec, prompts, seeds, device)
        out["other_configs"] = extras
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, state, codec_state)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

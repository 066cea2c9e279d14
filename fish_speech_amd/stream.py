"""Streaming chunked decode (BASELINE.json config 5): the Dual-AR frame loop interleaved with an
incremental codec decode, so the first audio leaves after ``first_chunk_frames`` frames instead of
after the whole utterance.

The reference streams at text-chunk granularity only: ``generate_long`` yields one
``GenerateResponse`` of codes per text chunk (fish_speech/models/text2semantic/inference.py:690-733)
and the engine decodes each with ``decode_vq_tokens`` and emits ``InferenceResult("segment")``
(fish_speech/inference_engine/__init__.py:73-140).  Here the same two calls are cut at frame
granularity.  What is emitted is exactly what the offline path produces:

* codes: the frames ``generate_batch`` yields (same slots, same graph-replayed kernels);
* audio: every codec layer is causal (modded_dac.py:521-588, window mask 380-398), so the samples of
  frames [t0, t1) do not depend on later frames; ``MiDAC.from_indices_tail`` computes them from the new
  codes and the state it kept of the earlier frames (per-layer K/V of the windowed transformer, its output, the
  upsampled latents; the decoder conv stack re-reads 5 frames of left context) and the concatenation of all chunks
  is bit-identical to ``from_indices`` over the final codes (tests/test_stream_gpu.py).

Like ``generate_long`` (inference.py:708, ``codes = y[1:, T:-1]``) the last generated frame of an
utterance -- the ``<|im_end|>`` frame, or the one that hit ``max_new_tokens`` -- is never voiced.  A live utterance's
newest frame is known not to be that frame as soon as the poll after it says "not ended" and the frame budget is not
used up (another frame WILL follow), so it is voiced right away: the first chunk leaves after ``first_chunk_frames``
generated frames, not one later.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Iterator, List, Optional, Sequence

import torch

from .dac import MiDAC
from .dual_ar import MiDualAR


@dataclass
class StreamChunk:
    t0: int                 # first frame of this chunk
    t1: int                 # one past the last frame
    audio: torch.Tensor     # (B, 1, (t1-t0)*frame_length) fp32 on the device
    codes: torch.Tensor     # (B, num_codebooks, t1-t0) int64 on the device
    valid_frames: List[int]  # per utterance: how many of the chunk's frames belong to it (0 once it ended)
    finished: List[bool]    # per utterance: generation has ended


def chunk_schedule(total_frames: int, first_chunk_frames: int, chunk_frames: int, growth: float = 1.0,
                   max_chunk_frames: int = 256) -> List[int]:
    """Frame counts at which audio is emitted: first_chunk_frames, then every chunk_frames.  growth > 1 lengthens every
    later chunk by that factor (up to max_chunk_frames): frames are generated several times faster than they play, so
    the playback buffer grows and later chunks can be longer -- fewer incremental codec decodes, each of which re-reads
    the decoder's left context and runs latency-bound small launches."""
    if first_chunk_frames < 1 or chunk_frames < 1:
        raise ValueError("chunk sizes must be >= 1")
    if growth < 1.0 or max_chunk_frames < 1:
        raise ValueError("growth must be >= 1 and max_chunk_frames >= 1")
    marks, t, step = [], min(first_chunk_frames, total_frames), float(min(chunk_frames, max_chunk_frames))
    while t < total_frames:
        marks.append(t)
        t += max(1, int(step))
        step = min(step * growth, float(max_chunk_frames))
    marks.append(total_frames)
    return marks


@torch.no_grad()
def generate_stream(*, model: MiDualAR, codec: MiDAC, prompts: Sequence[torch.Tensor], max_new_tokens: int,
                    first_chunk_frames: int = 8, chunk_frames: int = 32, seeds: Optional[Sequence[int]] = None,
                    stop_on_im_end: bool = True, temperature: float = 1.0, top_p: float = 0.9, top_k: int = 30,
                    use_ras: bool = True, timing: Optional[list] = None, chunk_growth: float = 1.0,
                    max_chunk_frames: int = 256, reuse_prefix: bool = False) -> Iterator[StreamChunk]:
    """Generate a batch of utterances and yield their audio chunk by chunk.

    Utterance i's audio is the concatenation over chunks of ``chunk.audio[i, :, :valid_frames[i]*frame_length]``
    and equals ``codec.from_indices(generate_batch(...)[i][1:, T_i:-1])``.  ``timing``: a list that receives the host
    wall time of every phase per chunk (tools/stream_latency.py --timing).  ``reuse_prefix`` (one prompt only): keep
    the slot's K/V when the stream ends and, next time, prefill only the columns past the longest shared prefix
    (``MiDualAR.prefill``); the caller releases slot 0 when the conversation is over."""
    cfg = model.config
    n = len(prompts)
    for p in prompts:
        if p.size(1) >= cfg.max_seq_len:
            raise ValueError(f"Input sequence length {p.size(1)} exceeds max_seq_len {cfg.max_seq_len}")
    if not model._cache_setup_done:
        model.setup_caches(max_batch_size=max(n, 1), max_seq_len=cfg.max_seq_len)
    if n > model.max_batch_size:
        raise ValueError(f"batch {n} exceeds max_batch_size {model.max_batch_size}")
    mn = [min(max_new_tokens if max_new_tokens else cfg.max_seq_len - p.size(1), cfg.max_seq_len - p.size(1))
          for p in prompts]
    slots = list(range(n))
    seeds = list(seeds) if seeds is not None else [model.next_seed() for _ in range(n)]
    samp = [model._sampling(temperature, top_p, top_k, seeds[i], use_ras) for i in range(n)]
    reuse_prefix = reuse_prefix and n == 1
    if reuse_prefix:
        model.prefill(slots, prompts, mn, samp, reuse_prefix=True)   # generates frame 0 of every utterance
    else:
        model.prefill(slots, prompts, mn, samp)
    total = max(mn)
    generated = 1
    emitted = 0
    stream_id = codec.new_stream_id()                  # the codec keeps its quantizer-side state between our calls
    length = [None] * n                                # final frame count of an utterance once it ended
    try:
        for mark in chunk_schedule(total, first_chunk_frames, chunk_frames, chunk_growth, max_chunk_frames):
            tm = [time.perf_counter()] if timing is not None else None
            if mark > generated:
                model.decode(slots, mark - generated)
                generated = mark
            if tm: tm.append(time.perf_counter())
            done = model.poll_done(slots) if stop_on_im_end else [0] * n
            if tm: tm.append(time.perf_counter())
            for i in slots:
                if length[i] is None and (done[i] or generated >= mn[i]):
                    length[i] = model.read(i)[0].shape[0] if done[i] else mn[i]
            finished = [length[i] is not None for i in slots]
            # voiced frames of utterance i: [0, length-1) once it ended; a live one (not ended by the poll above, budget
            # not used up) will get another frame, so all of its frames so far are voiceable
            voiced = [length[i] - 1 if finished[i] else generated for i in slots]
            t1 = max(voiced)
            if t1 > emitted:
                frames = model.frames_device(n, t1)                       # (B, t1, 1+ncb) int32
                codes = frames[:, :, 1:].permute(0, 2, 1).to(torch.int64).contiguous()
                if tm: tm.append(time.perf_counter())
                audio = codec.from_indices_tail(codes, emitted, stream_id=stream_id)
                if tm:
                    tm.append(time.perf_counter())
                    timing.append(dict(t1=t1, decode_call=tm[1] - tm[0], poll=tm[2] - tm[1], frames=tm[3] - tm[2],
                                       codec_call=tm[4] - tm[3]))
                yield StreamChunk(emitted, t1, audio, codes[:, :, emitted:t1],
                                  [max(0, min(v, t1) - emitted) for v in voiced], finished)
                emitted = t1
            if all(finished):
                break
    finally:
        # the codec's incremental state of this stream is void now: closed, its buffers serve the next stream (left to
        # age out of the 16 kept states, every new stream allocated ~1 GB of state before its first chunk)
        close = getattr(codec, "close_stream", None)
        if close is not None:
            close(stream_id)
        if not reuse_prefix:
            for i in slots:
                model.release(i)

"""Engine-level streaming (reference: fish_speech/inference_engine/__init__.py:28-140 `TTSInferenceEngine.inference`,
utils.py `InferenceResult` / `wav_chunk_header`, tools/server/inference.py:12-45).

The reference engine streams at TEXT-chunk granularity: one `InferenceResult("segment")` per `GenerateResponse` of
`generate_long`, each decoded whole by `decode_vq_tokens`.  `StreamingTTSEngine.inference` keeps that protocol --
"header" (wav header bytes) when streaming, "segment"s, one "final" with the whole utterance, "error" on failure --
but a segment leaves every `chunk_frames` frames (after `first_chunk_frames` for the first one): the Dual-AR frame
loop and the incremental codec decode of `stream.generate_stream` are interleaved, so first audio needs
`first_chunk_frames` frames instead of a whole text chunk.  The concatenation of the segments IS the final audio, and
equals `from_indices` over the codes the offline path generates (tests/test_stream_gpu.py)."""
from __future__ import annotations

import contextlib
import io
import wave
from dataclasses import dataclass
from typing import Iterator, List, Literal, Optional, Sequence, Tuple

import numpy as np
import torch

from .prompt import Conversation, Message, TextPart, VQPart
from .reference_loader import ReferenceLoader, VQManager
from .stream import generate_stream
from .text2semantic import _system_message, group_turns_into_batches, split_text_by_speaker

AMPLITUDE = 32768   # tools/server/inference.py:9


@dataclass
class InferenceResult:                       # inference_engine/utils.py:9-13
    code: Literal["header", "segment", "error", "final"]
    audio: Optional[Tuple[int, np.ndarray]]
    error: Optional[Exception]


def wav_chunk_header(sample_rate: int = 44100, bit_depth: int = 16, channels: int = 1) -> bytes:
    """An empty wav file = the header a streaming client prepends (inference_engine/utils.py:16-29)."""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as f:
        f.setnchannels(channels)
        f.setsampwidth(bit_depth // 8)
        f.setframerate(sample_rate)
    return buf.getvalue()


@dataclass
class TTSRequest:
    """The fields of ServeTTSRequest (fish_speech/utils/schema.py) the engine reads."""
    text: str
    streaming: bool = False
    max_new_tokens: int = 1024
    top_p: float = 0.8
    temperature: float = 0.8
    chunk_length: int = 200
    seed: Optional[int] = None
    reference_id: Optional[str] = None              # references/<id>/*.wav + .lab (ReferenceLoader.load_by_id)
    references: Sequence = ()                       # ServeReferenceAudio-like objects: .audio (wav bytes), .text
    use_memory_cache: Literal["on", "off"] = "off"
    prompt_texts: Sequence[str] = ()                # already-encoded references (extension; used when neither of
    prompt_tokens: Sequence[torch.Tensor] = ()      # the two above is given): texts + (num_codebooks, n) codes
    top_k: int = 30
    first_chunk_frames: int = 8
    chunk_frames: int = 32
    chunk_growth: float = 2.0      # later chunks 32, 64, 128 ... frames: generation outruns playback, fewer codec calls
    max_chunk_frames: int = 256


class StreamingTTSEngine(ReferenceLoader, VQManager):
    """TTSInferenceEngine(ReferenceLoader, VQManager) (inference_engine/__init__.py:22-37) over MiDualAR + MiDAC."""

    def __init__(self, model, codec, precision=torch.bfloat16):
        super().__init__()
        self.model, self.decoder_model, self.precision = model, codec, precision
        self.lock_timeout = 600.0     # seconds a request waits for the model before it reports an error

    @torch.no_grad()
    def inference(self, req: TTSRequest) -> Iterator[InferenceResult]:
        """Threading contract: may be called from any number of request threads (the reference's api_server does); the
        model's lock is held from the first `next()` until the generator finishes or is closed, so requests are served
        one after the other -- the reference serialises them through its single-worker llama queue.  A consumer that
        abandons the generator must `close()` it (a `for` loop that breaks does)."""
        model, codec = self.model, self.decoder_model
        lock = getattr(model, "lock", None)
        # explicit acquire / release (not `with`): the lock is a plain threading.Lock, so whichever thread runs the
        # generator's last step (next(), close() or finalisation) may release it, and a second request started on the
        # SAME thread while this one is suspended waits (and reports a timeout) instead of sharing slot 0
        if lock is not None and not lock.acquire(timeout=self.lock_timeout):
            yield InferenceResult("error", None, TimeoutError(
                f"the model stayed busy for {self.lock_timeout:.0f} s (another request holds it; a suspended generator on "
                "this very thread would never release it)"))
            return
        try:
            yield from self._inference_locked(req, model, codec)
        finally:
            if lock is not None:
                lock.release()

    def _inference_locked(self, req: TTSRequest, model, codec) -> Iterator[InferenceResult]:
        sample_rate = codec.sample_rate
        try:
            if req.streaming:
                yield InferenceResult("header", (sample_rate, np.array(wav_chunk_header(sample_rate=sample_rate))), None)
            # references by id or by content hash, encoded by the codec and cached (inference_engine/__init__.py:47-56)
            prompt_tokens, prompt_texts = list(req.prompt_tokens), list(req.prompt_texts)
            if req.reference_id is not None:
                prompt_tokens, prompt_texts = self.load_by_id(req.reference_id, req.use_memory_cache)
            elif req.references:
                prompt_tokens, prompt_texts = self.load_by_hash(list(req.references), req.use_memory_cache)
            use_prompt = bool(prompt_texts) and bool(prompt_tokens)
            system = _system_message(list(prompt_texts) if use_prompt else None,
                                     [c.cpu() for c in prompt_tokens] if use_prompt else None)
            turns = split_text_by_speaker(req.text)
            chunks = group_turns_into_batches(turns, max_speakers=5, max_bytes=req.chunk_length) if turns else [req.text]
            history = Conversation([system])
            segments: List[np.ndarray] = []
            for ci, chunk in enumerate(chunks):
                history.append(Message(role="user", parts=[TextPart(text=chunk)]))
                asking = history.copy()
                asking.append(Message(role="assistant", parts=[], modality="voice", add_im_end=False))
                prompt, _, _ = asking.encode_for_inference(model.tokenizer, num_codebooks=model.config.num_codebooks)
                if prompt.size(1) > model.config.max_seq_len - 2048:    # text2semantic/inference.py:658-661
                    raise ValueError(f"Prompt is too long: {prompt.size(1)} > {model.config.max_seq_len - 2048}")
                seeds = None if req.seed is None else [int(req.seed) + ci]
                codes_parts = []
                # the engine decodes under autocast(self.precision) (inference_engine/__init__.py:183-186)
                with torch.autocast("cuda", dtype=self.precision, enabled=self.precision is not None):
                    for ch in generate_stream(model=model, codec=codec, prompts=[prompt],
                                              max_new_tokens=req.max_new_tokens, first_chunk_frames=req.first_chunk_frames,
                                              chunk_frames=req.chunk_frames, chunk_growth=req.chunk_growth,
                                              max_chunk_frames=req.max_chunk_frames, seeds=seeds, temperature=req.temperature,
                                              top_p=req.top_p, top_k=req.top_k, reuse_prefix=True):
                        n = ch.valid_frames[0]
                        if n <= 0:
                            continue
                        seg = ch.audio[0, 0, : n * codec.frame_length].float().cpu().numpy()
                        codes_parts.append(ch.codes[0, :, :n].cpu())
                        segments.append(seg)
                        if req.streaming:
                            yield InferenceResult("segment", (sample_rate, seg), None)
                if codes_parts:
                    history.append(Message(role="assistant", parts=[VQPart(codes=torch.cat(codes_parts, dim=1))],
                                           modality="voice"))
            if not segments:
                yield InferenceResult("error", None, RuntimeError("No audio generated, please check the input text."))
            else:
                yield InferenceResult("final", (sample_rate, np.concatenate(segments, axis=0)), None)
        except Exception as e:   # the reference reports worker errors as a result, not as a raise (__init__.py:88-98)
            yield InferenceResult("error", None, e)
        finally:
            # every text chunk's prompt repeats the conversation so far: the slot's prefill K/V were kept between
            # the chunks (prefix-KV reuse, like text2semantic.generate_long) and are dropped with the request
            try:
                model.release(0)
            except Exception:      # noqa: BLE001 -- nothing was prefilled (the request failed before its first chunk)
                pass


class BatchingTTSEngine(StreamingTTSEngine):
    """The same `inference()` protocol, but concurrent callers SHARE the GPU: every request thread's utterances go
    through one `serving.serve_stream` loop (continuous batching, up to `max_batch` utterances advance per pass over the
    weights, each streamed on its own chunk schedule) instead of taking turns behind the model's lock.  The reference
    serves one request at a time (its single-worker llama queue, text2semantic/inference.py:748-799); a request's
    audio here is still exactly what it would be alone -- the kernels are batch-invariant (tests/test_stream_gpu.py).

    One daemon thread runs the loop while there is work and holds `model.lock` only then, so other users of the
    model get it between bursts.  Text chunks of one request are generated one after the other (each chunk's prompt
    contains the previous chunk's codes); prefix-KV reuse across chunks is not used on this path."""

    def __init__(self, model, codec, precision=torch.bfloat16, max_batch: int = 8, step_frames: int = 8,
                 result_timeout: float = 600.0):
        super().__init__(model, codec, precision)
        import itertools
        import threading

        from .serving import RequestFeed

        self.max_batch, self.step_frames, self.result_timeout = max_batch, step_frames, result_timeout
        self._feed = RequestFeed()
        self._queues: dict = {}
        self._rid = itertools.count()
        self._mutex = threading.Lock()
        self._thread = None

    # ---- the serving thread
    def _ensure_thread(self):
        import threading

        with self._mutex:
            if self._thread is None or not self._thread.is_alive():
                self._thread = threading.Thread(target=self._serve, name="fishmi-serve", daemon=True)
                self._thread.start()

    def _serve(self):
        from .serving import serve_stream

        model, codec = self.model, self.decoder_model
        while not self._feed.closed:
            if not self._feed.wait(1.0):
                continue
            try:
                with getattr(model, "lock", None) or contextlib.nullcontext(), torch.no_grad(), \
                        torch.autocast("cuda", dtype=self.precision, enabled=self.precision is not None):
                    # (the serving thread acquires and releases on itself: `with` is fine here)
                    for ev in serve_stream(model=model, codec=codec, requests=self._feed, max_batch=self.max_batch,
                                           step_frames=self.step_frames, return_when_idle=True):
                        q = self._queues.get(ev.rid)
                        if q is not None:
                            q.put(ev)
            except Exception as e:   # noqa: BLE001 -- every waiting request gets the error instead of a hang
                for q in list(self._queues.values()):
                    q.put(e)

    def close(self):
        self._feed.close()

    # ---- one utterance through the shared loop
    def _stream_utterance(self, prompt, req: TTSRequest, seed):
        import queue

        from .serving import StreamRequest

        rid = next(self._rid)
        q: "queue.Queue" = queue.Queue()
        self._queues[rid] = q
        sreq = StreamRequest(prompt=prompt, max_new_tokens=req.max_new_tokens, seed=seed, rid=rid,
                             temperature=req.temperature, top_p=req.top_p, top_k=req.top_k,
                             first_chunk_frames=req.first_chunk_frames, chunk_frames=req.chunk_frames,
                             chunk_growth=req.chunk_growth, max_chunk_frames=req.max_chunk_frames)
        try:
            self._feed.put(sreq)
            self._ensure_thread()
            while True:
                ev = q.get(timeout=self.result_timeout)
                if isinstance(ev, Exception):
                    raise ev
                if ev.kind == "error":       # this request alone was refused at admission (serving.serve_stream)
                    raise ValueError(ev.error)
                if ev.kind == "final":
                    return
                yield ev
        finally:
            sreq.cancelled = True            # no-op once it ended; frees the slot if the consumer went away early
            self._queues.pop(rid, None)

    @torch.no_grad()
    def inference(self, req: TTSRequest) -> Iterator[InferenceResult]:
        model, codec = self.model, self.decoder_model
        sample_rate = codec.sample_rate
        try:
            if req.streaming:
                yield InferenceResult("header", (sample_rate, np.array(wav_chunk_header(sample_rate=sample_rate))), None)
            prompt_tokens, prompt_texts = list(req.prompt_tokens), list(req.prompt_texts)
            if req.reference_id is not None:
                prompt_tokens, prompt_texts = self.load_by_id(req.reference_id, req.use_memory_cache)
            elif req.references:
                prompt_tokens, prompt_texts = self.load_by_hash(list(req.references), req.use_memory_cache)
            use_prompt = bool(prompt_texts) and bool(prompt_tokens)
            system = _system_message(list(prompt_texts) if use_prompt else None,
                                     [c.cpu() for c in prompt_tokens] if use_prompt else None)
            turns = split_text_by_speaker(req.text)
            chunks = group_turns_into_batches(turns, max_speakers=5, max_bytes=req.chunk_length) if turns else [req.text]
            history = Conversation([system])
            segments: List[np.ndarray] = []
            for ci, chunk in enumerate(chunks):
                history.append(Message(role="user", parts=[TextPart(text=chunk)]))
                asking = history.copy()
                asking.append(Message(role="assistant", parts=[], modality="voice", add_im_end=False))
                prompt, _, _ = asking.encode_for_inference(model.tokenizer, num_codebooks=model.config.num_codebooks)
                if prompt.size(1) > model.config.max_seq_len - 2048:    # text2semantic/inference.py:658-661
                    raise ValueError(f"Prompt is too long: {prompt.size(1)} > {model.config.max_seq_len - 2048}")
                codes_parts = []
                for ev in self._stream_utterance(prompt, req, None if req.seed is None else int(req.seed) + ci):
                    seg = ev.audio[0, 0].float().cpu().numpy()
                    codes_parts.append(ev.codes.cpu())
                    segments.append(seg)
                    if req.streaming:
                        yield InferenceResult("segment", (sample_rate, seg), None)
                if codes_parts:
                    history.append(Message(role="assistant", parts=[VQPart(codes=torch.cat(codes_parts, dim=1))],
                                           modality="voice"))
            if not segments:
                yield InferenceResult("error", None, RuntimeError("No audio generated, please check the input text."))
            else:
                yield InferenceResult("final", (sample_rate, np.concatenate(segments, axis=0)), None)
        except Exception as e:   # the reference reports worker errors as a result, not as a raise (__init__.py:88-98)
            yield InferenceResult("error", None, e)


def inference_wrapper(req: TTSRequest, engine: StreamingTTSEngine):
    """tools/server/inference.py:12-45: the byte stream the HTTP endpoint sends."""
    count = 0
    for result in engine.inference(req):
        if result.code == "header":
            if isinstance(result.audio, tuple):
                yield result.audio[1]
        elif result.code == "error":
            raise RuntimeError(str(result.error))
        elif result.code == "segment":
            count += 1
            if isinstance(result.audio, tuple):
                yield (result.audio[1] * AMPLITUDE).astype(np.int16).tobytes()
        elif result.code == "final":
            count += 1
            if isinstance(result.audio, tuple):
                yield result.audio[1]
            return
    if count == 0:
        raise RuntimeError("No audio generated, please check the input text.")

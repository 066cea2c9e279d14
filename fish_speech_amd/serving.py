"""One engine loop that streams AND refills slots (BASELINE.json configs 4 + 5 together).

The reference serves one request at a time (fish_speech/inference_engine/__init__.py:73-140 pulls one
`GenerateRequest` through its single-worker llama queue, text2semantic/inference.py:748-799) and emits audio per text
chunk.  `scheduler.generate_queue` adds continuous batching (finished slots are refilled), `stream.generate_stream`
adds frame-granular audio -- but only for utterances started together.  `serve_stream` combines them: up to
`max_batch` utterances are in flight, every one on its OWN chunk schedule counted from its own first frame, a finished
utterance's slot is refilled at the next poll, and every utterance's audio equals what the offline path
(`generate` + `from_indices`) produces for it, bit for bit (tests/test_stream_gpu.py).

Mechanics: all live slots advance together by `step_frames` frames per `MiDualAR.decode` call (one hipGraph replay
per frame); after each advance the loop decodes, for every utterance whose schedule is due, the frames it has not
voiced yet with `MiDAC.from_indices_tail(stream_id=...)` -- the codec keeps one quantizer-side state per open stream
(csrc/dac.hip: select_stream_state).  Like `generate_long` (inference.py:708) the last generated frame of an
utterance is never voiced (see stream.py for why a live utterance's newest frame can be)."""
from __future__ import annotations

import collections
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Iterable, Iterator, List, Optional

import torch

from .stream import chunk_schedule


@dataclass
class StreamRequest:
    prompt: torch.Tensor                     # (1+ncb, T) like the reference's `generate`
    max_new_tokens: int = 0                  # 0 = up to max_seq_len
    seed: Optional[int] = None
    rid: int = -1                            # caller's id, echoed in the events
    arrival: float = 0.0                     # seconds after the loop starts at which the request exists
    # per-request overrides of the loop's defaults (None = the loop's): sampling and chunk schedule
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    first_chunk_frames: Optional[int] = None
    chunk_frames: Optional[int] = None
    chunk_growth: Optional[float] = None
    max_chunk_frames: Optional[int] = None
    cancelled: bool = False                  # set by the owner at any time: the utterance ends at the next poll
    arrival_abs: Optional[float] = None      # stamped by RequestFeed.put (its clock); arrival = arrival_abs - loop start


class RequestFeed:
    """Thread-safe, open-ended source of StreamRequests for a long-running `serve_stream` loop: producers `put`
    requests at any time, the loop `poll`s without blocking while utterances are live and `wait`s when idle."""

    def __init__(self, clock: Callable[[], float] = time.perf_counter):
        self._q: collections.deque = collections.deque()
        self._cv = threading.Condition()
        self._clock = clock
        self.closed = False

    def put(self, req: StreamRequest) -> None:
        with self._cv:
            if self.closed:
                raise RuntimeError("RequestFeed is closed")
            req.arrival_abs = self._clock()
            self._q.append(req)
            self._cv.notify_all()

    def poll(self) -> Optional[StreamRequest]:
        with self._cv:
            return self._q.popleft() if self._q else None

    def wait(self, timeout: float) -> bool:
        """Block until a request is queued or the feed is closed (or the timeout passes); True if one is queued."""
        with self._cv:
            if not self._q and not self.closed:
                self._cv.wait(timeout)
            return bool(self._q)

    def close(self) -> None:
        with self._cv:
            self.closed = True
            self._cv.notify_all()

    def __len__(self) -> int:
        with self._cv:
            return len(self._q)


@dataclass
class StreamEvent:
    rid: int
    kind: str                                # "segment" | "final" | "error"
    t0: int                                  # first frame of the segment
    t1: int                                  # one past its last frame
    audio: Optional[torch.Tensor]            # (1, 1, (t1 - t0) * frame_length) fp32 on the device ("segment")
    codes: Optional[torch.Tensor]            # (num_codebooks, t1 - t0) int64 on the device ("segment")
    t_emit: float                            # seconds since the loop started
    first_audio_latency: Optional[float] = None   # on an utterance's first segment: t_emit - max(arrival, 0)
    error: Optional[str] = None                   # kind "error": why this request was not admitted (feed-driven loops only)


@dataclass
class _Live:
    req: StreamRequest
    slot: int
    limit: int                               # frames this utterance may generate
    marks: List[int]                         # generated-frame counts at which audio is due
    generated: int = 1                       # frames generated so far (the prefill makes frame 0)
    emitted: int = 0                         # frames voiced so far
    stream_id: int = 0
    length: Optional[int] = None             # final frame count once the utterance ended
    first_done: bool = False
    events: list = field(default_factory=list)


@torch.no_grad()
def serve_stream(*, model, codec, requests: Iterable[StreamRequest], max_batch: Optional[int] = None,
                 step_frames: int = 8, first_chunk_frames: int = 8, chunk_frames: int = 32, chunk_growth: float = 2.0,
                 max_chunk_frames: int = 256, temperature: float = 1.0, top_p: float = 0.9, top_k: int = 30,
                 use_ras: bool = True, clock: Callable[[], float] = time.perf_counter,
                 wait: Callable[[float], None] = time.sleep, admit_early: bool = True,
                 return_when_idle: bool = False, open_slot_step: int = 2) -> Iterator[StreamEvent]:
    """Serve `requests` (any iterable, consumed lazily in order, or a `RequestFeed` that other threads fill while the
    loop runs) through `max_batch` slots; yields StreamEvents as audio becomes available.  A request is admitted once
    `clock() - start >= request.arrival` and a slot is free.  With a feed nothing is known about future arrivals, so
    while a slot is free the live utterances advance `open_slot_step` frames at a time (a newcomer waits for at most
    that); `return_when_idle` ends the loop when nothing is live and the feed is empty instead of waiting for it.
    `admit_early`: while a slot is free and the next request is known but not yet due, the advance is cut short so
    that it ends about when the request arrives (frame time measured on the fly) -- a new utterance then waits for
    the rest of ONE frame instead of the rest of a `step_frames` advance; nothing about any utterance's audio
    depends on how the advances are cut."""
    cfg = model.config
    if not model._cache_setup_done:
        model.setup_caches(max_batch_size=max_batch or 8, max_seq_len=cfg.max_seq_len)
    B = min(max_batch or model.max_batch_size, model.max_batch_size)
    if step_frames < 1:
        raise ValueError("step_frames must be >= 1")
    feed = requests if hasattr(requests, "poll") else None
    it = None if feed is not None else iter(requests)
    pending: Optional[StreamRequest] = None
    exhausted = False
    free = list(range(B - 1, -1, -1))
    live: dict = {}
    start = clock()
    fl = codec.frame_length
    frame_s = 0.0                            # measured wall time per decode frame (moving average)

    def next_request():
        nonlocal pending, exhausted
        if pending is None and feed is not None:
            pending = feed.poll()
            if pending is not None:          # it exists NOW: arrival = when it was put, for the latency report
                pending.arrival = min((pending.arrival_abs if pending.arrival_abs is not None else clock()) - start,
                                      clock() - start)
        elif pending is None and not exhausted:
            try:
                pending = next(it)
            except StopIteration:
                exhausted = True
        return pending

    def opt(v, default):
        return default if v is None else v

    try:
        while True:
            # ---- admission: every arrived request that finds a free slot is prefilled in ONE call
            new: List[_Live] = []
            samp = []
            while free and next_request() is not None and clock() - start >= pending.arrival:
                r, pending = pending, None
                # a request is validated BEFORE it joins the prefill call.  From a list of requests a bad one raises (the
                # caller's bug); from a feed it fails alone -- an "error" event for its owner, the loop and everybody
                # else's utterances go on (one bad request used to end the loop for all of them)
                try:
                    if not torch.is_tensor(r.prompt):
                        raise TypeError(f"prompt must be an integer tensor ({cfg.num_codebooks + 1}, T), got {type(r.prompt).__name__}")
                    if r.prompt.dim() != 2 or r.prompt.size(0) != cfg.num_codebooks + 1:
                        raise ValueError(f"prompt must be ({cfg.num_codebooks + 1}, T), got {tuple(r.prompt.shape)}")
                    T = r.prompt.size(1)
                    if T >= cfg.max_seq_len:  # inference.py:263-266
                        raise ValueError(f"Input sequence length {T} exceeds max_seq_len {cfg.max_seq_len}")
                    limit = min(r.max_new_tokens if r.max_new_tokens else cfg.max_seq_len - T, cfg.max_seq_len - T)
                    marks = chunk_schedule(limit, opt(r.first_chunk_frames, first_chunk_frames), opt(r.chunk_frames, chunk_frames),
                                           opt(r.chunk_growth, chunk_growth), opt(r.max_chunk_frames, max_chunk_frames))
                    sp = model._sampling(opt(r.temperature, temperature), opt(r.top_p, top_p), opt(r.top_k, top_k),
                                         r.seed if r.seed is not None else model.next_seed(), use_ras)
                except Exception as e:   # noqa: BLE001 -- whatever a malformed request trips, it fails ALONE
                    if feed is None:
                        raise
                    yield StreamEvent(r.rid, "error", 0, 0, None, None, clock() - start, error=str(e))
                    continue
                new.append(_Live(r, free.pop(), limit, marks, stream_id=codec.new_stream_id()))
                samp.append(sp)
            if new:
                model.prefill([u.slot for u in new], [u.req.prompt for u in new], [u.limit for u in new], samp)
                live.update({u.slot: u for u in new})
            if not live:
                if next_request() is None:
                    if feed is None or feed.closed or return_when_idle:
                        return
                    feed.wait(0.05)                                      # idle until somebody puts a request
                    continue
                wait(max(0.0, pending.arrival - (clock() - start)))      # idle until the next arrival
                continue
            # ---- cancelled utterances leave before the advance (their slots refill at the next admission)
            for s_ in [s_ for s_, u in live.items() if u.req.cancelled]:
                u = live.pop(s_)
                yield StreamEvent(u.req.rid, "final", 0, u.emitted, None, None, clock() - start)
                model.release(s_)
                _close_stream(codec, u.stream_id)
                free.append(s_)
            if not live:
                continue
            # ---- advance every live utterance
            slots = sorted(live)
            need = min(step_frames, max(u.limit - u.generated for u in live.values()))
            # an utterance that has not been heard yet: stop the advance at its first mark, not up to step_frames - 1 later
            for u in live.values():
                if not u.first_done and u.marks and u.marks[0] > u.generated:
                    need = max(1, min(need, u.marks[0] - u.generated))
            if feed is not None and free:
                need = max(1, min(need, open_slot_step))
            if feed is None and admit_early and need > 1 and free and frame_s > 0.0 and next_request() is not None:
                until = pending.arrival - (clock() - start)          # > 0: the admission loop above did not take it
                need = max(1, min(need, int(until / frame_s) + 1))
            t_adv = clock()
            if need > 0:
                model.decode(slots, need)
            done = model.poll_done(slots)
            if need > 0:
                dt = (clock() - t_adv) / need
                frame_s = dt if frame_s == 0.0 else 0.8 * frame_s + 0.2 * dt
            for s, d in zip(slots, done):
                u = live[s]
                u.generated = min(u.limit, u.generated + need)
                if u.length is None and (d or u.generated >= u.limit):
                    u.length = model.read(s)[0].shape[0] if d else u.limit
            # ---- voice what is due: the last frame of an utterance is never voiced; a live utterance (the poll above says
            # it has not ended and its budget is not used up) will get another frame, so its newest one is voiceable
            for s in slots:
                u = live[s]
                ended = u.length is not None
                voiced = u.length - 1 if ended else u.generated
                due = ended or any(u.emitted < m <= voiced for m in u.marks)
                if due and voiced > u.emitted:
                    frames = model.frames_device(model.max_batch_size, voiced)[s]          # (voiced, 1+ncb) int32
                    codes = frames[:, 1:].t().to(torch.int64).contiguous()                  # (ncb, voiced)
                    audio = codec.from_indices_tail(codes[None], u.emitted, stream_id=u.stream_id)
                    now = clock() - start
                    ev = StreamEvent(u.req.rid, "segment", u.emitted, voiced, audio, codes[:, u.emitted:voiced], now)
                    if not u.first_done:
                        ev.first_audio_latency = now - max(u.req.arrival, 0.0)
                        u.first_done = True
                    u.emitted = voiced
                    yield ev
                if ended:
                    yield StreamEvent(u.req.rid, "final", 0, u.emitted, None, None, clock() - start)
                    model.release(s)
                    _close_stream(codec, u.stream_id)
                    del live[s]
                    free.append(s)

    finally:                                 # an exception / a closed generator must not leak slots
        for s_ in list(live):
            try:
                model.release(s_)
                _close_stream(codec, live[s_].stream_id)
            except Exception:                # noqa: BLE001
                pass
            live.pop(s_, None)

def _close_stream(codec, stream_id) -> None:
    """An utterance's codec stream has ended: its incremental state goes now (MiDAC.close_stream), not by LRU ageing."""
    close = getattr(codec, "close_stream", None)
    if close is not None and stream_id:
        close(stream_id)


def collect(events: Iterable[StreamEvent], frame_length: int):
    """Concatenate every utterance's segments: {rid: (audio (n * frame_length,) fp32 CPU, codes (ncb, n) CPU)}."""
    parts: dict = {}
    for ev in events:
        if ev.kind == "segment":
            a, c = parts.setdefault(ev.rid, ([], []))
            a.append(ev.audio[0, 0].float().cpu())
            c.append(ev.codes.cpu())
    return {rid: (torch.cat(a) if a else torch.zeros(0), torch.cat(c, dim=1) if c else None) for rid, (a, c) in parts.items()}

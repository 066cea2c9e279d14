"""Host-side logic above the C ABI on the CPU, with stand-ins for the two device objects: the streaming schedule
(stream.generate_stream), the engine's InferenceResult protocol (engine.py; reference:
fish_speech/inference_engine/__init__.py:73-140, utils.py), the HTTP byte stream (tools/server/inference.py:12-45) and
the server's batch codec helpers (tools/server/model_utils.py:15-86).  The stand-ins implement exactly the slot API /
codec API those modules call; the GPU runs of the same code are in tests/test_stream_gpu.py."""
import io
import itertools
import time
import wave
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from fish_speech_amd import server_utils
from fish_speech_amd.engine import InferenceResult, StreamingTTSEngine, TTSRequest, inference_wrapper, wav_chunk_header
from fish_speech_amd.stream import chunk_schedule, generate_stream
from oracle.fake_tokenizer import ByteTokenizer

NCB = 9


class StubDualAR:
    """MiDualAR's slot API with a deterministic generator: utterance with seed s has 4 + (13 s) % 37 frames (its last
    one is the <|im_end|> frame) unless max_new_tokens cuts it; frame f holds codes (7 s + 3 f + c) % 1024."""

    def __init__(self, max_batch=4, max_seq_len=4096):
        self.config = SimpleNamespace(num_codebooks=NCB, max_seq_len=max_seq_len)
        self.max_batch_size, self._cache_setup_done = max_batch, True
        self.tokenizer = ByteTokenizer()
        self.slots, self._seed = {}, itertools.count(100)
        self.released = []

    def next_seed(self):
        return next(self._seed)

    def _sampling(self, t, p, k, seed, ras):
        return seed

    @staticmethod
    def natural_len(seed):
        return 4 + (13 * seed) % 37

    def prefill(self, slots, prompts, max_new, samp, reuse_prefix=False):
        for s, m, seed in zip(slots, max_new, samp):
            if reuse_prefix:                     # the slot was kept by the previous chunk of the conversation
                self.slots.pop(s, None)
            assert s not in self.slots
            self.slots[s] = dict(seed=seed, limit=min(m, self.natural_len(seed)), eos=self.natural_len(seed) <= m, n=1)

    def decode(self, slots, n_frames):
        for s in slots:
            st = self.slots[s]
            st["n"] = min(st["limit"], st["n"] + n_frames)

    def poll_done(self, slots):
        return [1 if (self.slots[s]["eos"] and self.slots[s]["n"] >= self.slots[s]["limit"]) else 0 for s in slots]

    def _frames(self, seed, n):
        f = torch.arange(n).view(-1, 1)
        c = torch.arange(NCB + 1).view(1, -1)
        return ((7 * seed + 3 * f + c) % 1024).to(torch.int32)

    def read(self, slot):
        st = self.slots[slot]
        return self._frames(st["seed"], st["n"]), 1

    def frames_device(self, n_slots, n_frames):
        out = torch.zeros(n_slots, n_frames, NCB + 1, dtype=torch.int32)
        for s in range(n_slots):
            if s not in self.slots:
                continue
            k = min(n_frames, self.slots[s]["n"])
            out[s, :k] = self._frames(self.slots[s]["seed"], k)
        return out

    def release(self, slot):
        self.released.append(slot)
        del self.slots[slot]

    def expected_codes(self, seed, max_new):
        """what generate_long keeps of an utterance: codes[1:, T:-1] (the last generated frame is never voiced)"""
        n = min(max_new, self.natural_len(seed))
        return self._frames(seed, n)[: n - 1, 1:].t().to(torch.int64)


class StubCodec:
    """MiDAC's decode API with a causal stand-in: sample j of frame t = (sum of the frame's codes + 0.001 t) * 1e-4 + 1e-7 j."""
    frame_length, sample_rate, device = 16, 44100, torch.device("cpu")

    def __init__(self):
        self.calls = []
        self._ids = itertools.count(1)

    def new_stream_id(self):
        return next(self._ids)

    def from_indices(self, codes):
        B, nb, T = codes.shape
        assert nb == NCB + 1 or nb == NCB
        base = (codes.sum(1).double() + 0.001 * torch.arange(T).double()) * 1e-4          # (B, T)
        j = torch.arange(self.frame_length).double() * 1e-7
        return (base[:, :, None] + j).reshape(B, 1, T * self.frame_length).float()

    def from_indices_tail(self, codes, t0, stream_id=None):
        self.calls.append((codes.shape[-1], t0, stream_id))
        return self.from_indices(codes)[..., t0 * self.frame_length:]

    def encode(self, padded, audio_lengths=None):
        B, _, N = padded.shape
        T = -(-N // self.frame_length)
        feats = torch.zeros(B, NCB + 1, T, dtype=torch.int64)
        for b in range(B):
            feats[b] = (padded[b, 0, :: self.frame_length][:T] * 1000).round().long().clamp(0, 1023)[None]
        return feats, torch.tensor([-(-int(n) // self.frame_length) for n in audio_lengths])


def test_chunk_schedule():
    assert chunk_schedule(215, 8, 32) == [8, 40, 72, 104, 136, 168, 200, 215]
    assert chunk_schedule(5, 8, 32) == [5]
    assert chunk_schedule(8, 8, 32) == [8]
    assert chunk_schedule(9, 8, 1) == [8, 9]
    with pytest.raises(ValueError):
        chunk_schedule(10, 0, 4)
    with pytest.raises(ValueError):
        chunk_schedule(10, 4, 0)
    # growing chunks: 32, 64, 128, capped
    assert chunk_schedule(215, 8, 32, growth=2.0) == [8, 40, 104, 215]
    assert chunk_schedule(600, 8, 32, growth=2.0, max_chunk_frames=100) == [8, 40, 104, 204, 304, 404, 504, 600]
    assert chunk_schedule(50, 8, 4, growth=1.5) == [8, 12, 18, 27, 40, 50]
    with pytest.raises(ValueError):
        chunk_schedule(10, 4, 4, growth=0.5)


@pytest.mark.parametrize("first,chunk,growth", [(8, 32, 1.0), (1, 1, 1.0), (3, 5, 1.0), (64, 64, 1.0), (2, 2, 2.0)])
def test_generate_stream_emits_exactly_the_offline_frames(first, chunk, growth):
    """Ragged batch (utterances end at 17, 30, 6 and -- cut by max_new_tokens -- 20 frames): per utterance the
    concatenation of the valid parts of all chunks is from_indices over codes[1:, T:-1]; the newest frame of a live
    utterance is held back; chunks continue one codec stream (t0 of a call = T of the previous one, same id); every
    slot is released."""
    model, codec = StubDualAR(max_batch=4), StubCodec()
    seeds = [1, 2, 3, 4]           # natural lengths 17, 30, 6, 19
    assert [model.natural_len(s) for s in seeds] == [17, 30, 6, 19]
    prompts = [torch.zeros(NCB + 1, 5 + i, dtype=torch.int64) for i in range(4)]
    max_new = 20
    audio = [[] for _ in seeds]
    codes = [[] for _ in seeds]
    last_t1 = 0
    for ch in generate_stream(model=model, codec=codec, prompts=prompts, max_new_tokens=max_new,
                              first_chunk_frames=first, chunk_frames=chunk, seeds=seeds, chunk_growth=growth):
        assert ch.t0 == last_t1 and ch.t1 > ch.t0
        last_t1 = ch.t1
        for i, v in enumerate(ch.valid_frames):
            assert 0 <= v <= ch.t1 - ch.t0
            audio[i].append(ch.audio[i, :, : v * codec.frame_length])
            codes[i].append(ch.codes[i, :, :v])
    for i, s in enumerate(seeds):
        want_codes = model.expected_codes(s, max_new)
        got_codes = torch.cat(codes[i], dim=1)
        assert torch.equal(got_codes, want_codes), i
        want = codec.from_indices(want_codes[None])[0]
        assert torch.equal(torch.cat(audio[i], dim=-1), want), i
    assert sorted(model.released) == [0, 1, 2, 3] and not model.slots
    # one codec stream: same id, contiguous, starts at 0
    ids = {c[2] for c in codec.calls}
    assert len(ids) == 1 and None not in ids
    assert codec.calls[0][1] == 0
    for a, b in zip(codec.calls, codec.calls[1:]):
        assert b[1] == a[0]


def test_serve_stream_cuts_an_advance_short_for_a_request_about_to_arrive():
    """admit_early: with a free slot and the next request known but not yet due, the advance ends about when it
    arrives (measured frame time), so it is prefilled at most one frame late instead of up to a whole advance."""
    from fish_speech_amd.serving import StreamRequest, serve_stream

    def run(early):
        model, codec = StubDualAR(max_batch=3), StubCodec()
        now = [0.0]
        calls = []
        orig_decode, orig_prefill = model.decode, model.prefill

        def decode(slots, n):
            now[0] += 0.010 * n                       # 10 ms of fake time per frame
            calls.append(("decode", n, round(now[0], 3)))
            return orig_decode(slots, n)

        def prefill(slots, *a, **k):
            calls.append(("prefill", len(slots), round(now[0], 3)))
            return orig_prefill(slots, *a, **k)

        model.decode, model.prefill = decode, prefill
        reqs = [StreamRequest(prompt=torch.zeros(NCB + 1, 5, dtype=torch.int64), seed=11, rid=0, arrival=0.0),   # 35 frames
                StreamRequest(prompt=torch.zeros(NCB + 1, 6, dtype=torch.int64), seed=8, rid=1, arrival=0.205)]
        evs = list(serve_stream(model=model, codec=codec, requests=iter(reqs), max_batch=3, step_frames=8,
                                clock=lambda: now[0], wait=lambda dt: now.__setitem__(0, now[0] + dt), admit_early=early))
        second = [c for c in calls if c[0] == "prefill"][1]
        return second[2] - 0.205, calls, evs

    late_fixed, _, ev0 = run(False)
    late_early, calls, ev1 = run(True)
    assert late_fixed > 0.02 and 0.0 <= late_early <= 0.0101, (late_fixed, late_early)
    assert any(c[0] == "decode" and c[1] < 8 for c in calls)
    # the audio does not depend on how the advances are cut
    a = {(e.rid, e.t0): e for e in ev0 if e.kind == "segment"}
    cat = lambda evs, rid: torch.cat([e.codes for e in evs if e.kind == "segment" and e.rid == rid], dim=1)
    for rid in (0, 1):
        assert torch.equal(cat(ev0, rid), cat(ev1, rid))


@pytest.mark.parametrize("step,first,chunk,growth", [(8, 8, 32, 2.0), (1, 1, 1, 1.0), (5, 3, 4, 1.5), (16, 2, 64, 1.0)])
def test_serve_stream_refills_slots_and_every_utterance_equals_its_offline_result(step, first, chunk, growth):
    """serving.serve_stream = continuous batching + per-utterance chunk schedules: 9 requests (natural lengths 4..40
    frames, two cut by max_new_tokens) through 3 slots, arriving over time on a fake clock.  Every utterance's
    concatenated segments are exactly its offline codes / audio (last frame never voiced), first-audio latency is
    reported once per utterance, every slot is released, and each utterance decodes on its own codec stream id."""
    from fish_speech_amd.serving import StreamRequest, collect, serve_stream

    model, codec = StubDualAR(max_batch=3), StubCodec()
    seeds = [3, 11, 5, 8, 2, 30, 7, 19, 4]
    limits = [0, 0, 9, 0, 0, 12, 0, 0, 0]
    now = [0.0]
    reqs = [StreamRequest(prompt=torch.zeros(NCB + 1, 5 + i, dtype=torch.int64), max_new_tokens=limits[i], seed=s, rid=100 + i,
                          arrival=0.06 * i) for i, s in enumerate(seeds)]

    def clock():
        now[0] += 0.05          # every look at the clock costs 50 ms of fake time
        return now[0]

    evs = list(serve_stream(model=model, codec=codec, requests=iter(reqs), max_batch=3, step_frames=step,
                            first_chunk_frames=first, chunk_frames=chunk, chunk_growth=growth, clock=clock,
                            wait=lambda dt: now.__setitem__(0, now[0] + dt)))
    got = collect(evs, codec.frame_length)
    finals = [e.rid for e in evs if e.kind == "final"]
    assert sorted(finals) == [r.rid for r in reqs] and not model.slots and len(model.released) == 9 and len(set(model.released)) >= 2
    for r, seed, lim in zip(reqs, seeds, limits):
        want = model.expected_codes(seed, lim if lim else 10 ** 6)
        audio, codes = got.get(r.rid, (torch.zeros(0), None))
        if want.shape[1] == 0:
            assert codes is None or codes.shape[1] == 0
            continue
        assert torch.equal(codes, want), r.rid
        assert torch.equal(audio, codec.from_indices(want[None])[0, 0]), r.rid
        firsts = [e for e in evs if e.rid == r.rid and e.first_audio_latency is not None]
        assert len(firsts) == 1 and firsts[0].t0 == 0 and firsts[0].first_audio_latency > 0
        segs = [e for e in evs if e.rid == r.rid and e.kind == "segment"]
        assert [e.t0 for e in segs[1:]] == [e.t1 for e in segs[:-1]]                 # gapless, in order
    ids = {}
    for (T, t0, sid) in codec.calls:
        ids.setdefault(sid, []).append((t0, T))
    assert len(ids) == sum(1 for r in reqs if r.rid in got)                          # one codec stream per utterance
    assert all(v == sorted(v) and v[0][0] == 0 for v in ids.values())


def test_reference_loader_and_vq_manager_mirror_the_reference_semantics(tmp_path):
    """inference_engine/reference_loader.py:23-260 + vq_manager.py:16-53 over a codec object: references by id (a
    folder of wav + .lab pairs) and by content hash, both caches, id validation, PCM scaling, and the engine
    resolving `reference_id` / `references` the way TTSInferenceEngine.inference does."""
    import io
    from types import SimpleNamespace

    from scipy.io import wavfile

    from fish_speech_amd.engine import StreamingTTSEngine, TTSRequest
    from fish_speech_amd.reference_loader import ReferenceLoader

    class Codec(StubCodec):                      # the stub model counts NCB codebooks: encode as many rows
        def encode(self, padded, audio_lengths=None):
            feats, lens = super().encode(padded, audio_lengths)
            return feats[:, :NCB], lens

    codec = Codec()
    eng = StreamingTTSEngine(StubDualAR(max_batch=1), codec, precision=None)
    eng.references_root = tmp_path / "references"
    sr = codec.sample_rate
    t = np.arange(sr // 10) / sr
    a = (0.3 * np.sin(2 * np.pi * 200 * t)).astype(np.float32)
    src = tmp_path / "voice.wav"
    wavfile.write(str(src), sr // 2, a[::2].copy())                    # half the rate: load_audio resamples
    with pytest.raises(ValueError):
        eng.load_by_id("../etc", "off")
    # (the HTTP server's add / delete / list routes are control plane and not mirrored: the folder is laid out by hand)
    (eng.references_root / "alice").mkdir(parents=True)
    (eng.references_root / "alice" / "sample.wav").write_bytes(src.read_bytes())
    (eng.references_root / "alice" / "sample.lab").write_text("hello from alice", encoding="utf-8")
    assert not hasattr(eng, "add_reference") and not hasattr(eng, "delete_reference")
    toks, texts = eng.load_by_id("alice", "on")
    assert texts == ["hello from alice"] and len(toks) == 1 and toks[0].shape[0] == NCB
    n_frames = -(-(len(a[::2]) * 2) // codec.frame_length)
    assert abs(toks[0].shape[1] - n_frames) <= 1                         # resampled back to the codec's rate
    again, _ = eng.load_by_id("alice", "on")
    assert again[0] is toks[0]                                           # cache hit: the very same tensor
    fresh, _ = eng.load_by_id("alice", "off")
    assert fresh[0] is not toks[0] and torch.equal(fresh[0], toks[0])
    buf = io.BytesIO()
    wavfile.write(buf, sr, a)
    ref = SimpleNamespace(audio=buf.getvalue(), text="by hash")
    t1, x1 = eng.load_by_hash([ref, ref], "on")
    assert x1 == ["by hash", "by hash"] and t1[1] is t1[0] and len(eng.ref_by_hash) == 1
    with pytest.raises(ValueError):
        eng.load_audio(b"not a wav file at all", sr)
    assert eng.encode_reference(None, True) is None and eng.encode_reference(buf.getvalue(), False) is None
    wav = eng.decode_vq_tokens(t1[0])
    assert wav.dim() == 1 and wav.shape[0] == t1[0].shape[1] * codec.frame_length
    # the engine resolves the request's references through the loader before building the prompt
    res = list(eng.inference(TTSRequest(text="hi", reference_id="alice", use_memory_cache="on", max_new_tokens=8, seed=3)))
    assert res[-1].code == "final"
    res = list(eng.inference(TTSRequest(text="hi", references=[ref], max_new_tokens=8, seed=3)))
    assert res[-1].code == "final"
    assert isinstance(eng, ReferenceLoader)
    # integer PCM is scaled by 2^(bits-1) like torchaudio / soundfile (32768 for int16, not 32767); 8-bit PCM is
    # unsigned and re-centred: (x - 128) / 128
    b16 = io.BytesIO()
    wavfile.write(b16, sr, np.array([-32768, -16384, 0, 16384, 32767], dtype=np.int16))
    assert np.array_equal(eng.load_audio(b16.getvalue(), sr), np.array([-1.0, -0.5, 0.0, 0.5, 32767 / 32768], dtype=np.float32))
    b8 = io.BytesIO()
    wavfile.write(b8, sr, np.array([0, 64, 128, 192, 255], dtype=np.uint8))
    assert np.array_equal(eng.load_audio(b8.getvalue(), sr), np.array([-1.0, -0.5, 0.0, 0.5, 127 / 128], dtype=np.float32))
    # AIFF PCM (round 5; `.aiff / .aif / .aifc` are in AUDIO_EXTENSIONS upstream, torchaudio decodes them there): big-endian
    # 16-bit mono from bytes, 24-bit stereo from a file path (down-mixed), same scaling as the WAV path
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        import aifc
    p16 = tmp_path / "mono16.aif"
    with aifc.open(str(p16), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes(np.array([-32768, -16384, 0, 16384, 32767], dtype=">i2").tobytes())
    assert np.array_equal(eng.load_audio(p16.read_bytes(), sr), np.array([-1.0, -0.5, 0.0, 0.5, 32767 / 32768], dtype=np.float32))
    p24 = tmp_path / "stereo24.aiff"
    with aifc.open(str(p24), "wb") as f:
        f.setnchannels(2); f.setsampwidth(3); f.setframerate(sr)
        vals = [(-8388608, -8388608), (4194304, 0), (0, -4194304)]
        f.writeframes(b"".join(int(v).to_bytes(3, "big", signed=True) for pair in vals for v in pair))
    assert np.array_equal(eng.load_audio(str(p24), sr), np.array([-1.0, 0.25, -0.25], dtype=np.float32))


def test_generate_stream_argument_errors():
    model, codec = StubDualAR(max_batch=2, max_seq_len=64), StubCodec()
    p = torch.zeros(NCB + 1, 5, dtype=torch.int64)
    with pytest.raises(ValueError, match="exceeds max_seq_len"):
        list(generate_stream(model=model, codec=codec, prompts=[torch.zeros(NCB + 1, 64, dtype=torch.int64)], max_new_tokens=4))
    with pytest.raises(ValueError, match="exceeds max_batch_size"):
        list(generate_stream(model=model, codec=codec, prompts=[p, p, p], max_new_tokens=4))
    with pytest.raises(ValueError):
        list(generate_stream(model=model, codec=codec, prompts=[p], max_new_tokens=4, chunk_frames=0))
    assert not model.slots            # released even though the generator raised


def test_wav_chunk_header_is_an_empty_wav_file():
    h = wav_chunk_header(sample_rate=44100)
    with wave.open(io.BytesIO(h), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 44100, 0)
    assert len(h) == 44


def test_engine_protocol_and_byte_stream():
    """header (streaming only) -> segments -> final; the segments concatenate to the final audio, which is the codec's
    decode of the codes the offline path keeps; two text chunks give two generations whose second prompt carries the
    first one's codes; the HTTP wrapper scales segments to int16 bytes."""
    model, codec = StubDualAR(max_batch=1), StubCodec()
    eng = StreamingTTSEngine(model, codec, precision=None)
    req = TTSRequest(text="hello there", streaming=True, max_new_tokens=64, seed=3, first_chunk_frames=2, chunk_frames=3)
    res = list(eng.inference(req))
    assert [r.code for r in res[:1]] == ["header"] and res[-1].code == "final"
    assert all(r.code == "segment" for r in res[1:-1]) and len(res) >= 4
    assert bytes(res[0].audio[1].tobytes()) == wav_chunk_header(sample_rate=codec.sample_rate)
    segs = np.concatenate([r.audio[1] for r in res[1:-1]])
    assert np.array_equal(segs, res[-1].audio[1]) and res[-1].audio[0] == codec.sample_rate
    want = codec.from_indices(model.expected_codes(3, 64)[None])[0, 0].numpy()
    assert np.array_equal(res[-1].audio[1], want)
    # not streaming: only the final result
    res2 = list(eng.inference(TTSRequest(text="hello there", streaming=False, max_new_tokens=64, seed=3)))
    assert [r.code for r in res2] == ["final"] and np.array_equal(res2[0].audio[1], want)
    # byte stream of the HTTP endpoint
    chunks = list(inference_wrapper(req, eng))
    assert bytes(chunks[0].tobytes()) == wav_chunk_header(sample_rate=codec.sample_rate)
    pcm = b"".join(chunks[1:-1])
    assert pcm == (want * 32768).astype(np.int16).tobytes()
    assert np.array_equal(chunks[-1], want)


@pytest.mark.timeout(60)
def test_serve_stream_over_a_request_feed_per_request_parameters_and_cancellation():
    """serve_stream fed by a RequestFeed (requests appear while the loop runs): per-request sampling parameters and chunk
    schedule override the loop's, a cancelled utterance leaves at the next poll and its slot is reused, the loop
    returns when idle if asked to, and a closed feed ends a waiting loop."""
    import threading

    from fish_speech_amd.serving import RequestFeed, StreamRequest, collect, serve_stream

    model, codec = StubDualAR(max_batch=2), StubCodec()
    seen = []
    orig = model._sampling
    model._sampling = lambda t, p, k, seed, ras: (seen.append((t, p, k)), orig(t, p, k, seed, ras))[1]
    feed = RequestFeed()
    a = StreamRequest(prompt=torch.zeros(NCB + 1, 5, dtype=torch.int64), seed=11, rid=1, temperature=0.3, top_k=5,
                      first_chunk_frames=2, chunk_frames=2)
    b = StreamRequest(prompt=torch.zeros(NCB + 1, 6, dtype=torch.int64), seed=8, rid=2)
    c = StreamRequest(prompt=torch.zeros(NCB + 1, 7, dtype=torch.int64), seed=5, rid=3)
    feed.put(a); feed.put(b); feed.put(c)                       # c waits for a slot
    assert len(feed) == 3 and a.arrival_abs is not None
    evs = []
    for ev in serve_stream(model=model, codec=codec, requests=feed, max_batch=2, step_frames=4, first_chunk_frames=8,
                           chunk_frames=32, temperature=0.9, top_p=0.8, top_k=30, return_when_idle=True):
        evs.append(ev)
        if ev.rid == 2 and ev.kind == "segment":
            b.cancelled = True                                  # the consumer of utterance 2 goes away
    got = collect(evs, codec.frame_length)
    assert (0.3, 0.8, 5) in seen and (0.9, 0.8, 30) in seen    # request a's overrides, the loop's defaults for b / c
    assert torch.equal(got[1][1], model.expected_codes(11, 10 ** 6)) and torch.equal(got[3][1], model.expected_codes(5, 10 ** 6))
    assert got[2][1].shape[1] < model.expected_codes(8, 10 ** 6).shape[1]          # cut short, yet a final was sent
    assert sorted(e.rid for e in evs if e.kind == "final") == [1, 2, 3] and not model.slots
    segs_a = [e for e in evs if e.rid == 1 and e.kind == "segment"]
    assert segs_a[0].t1 == 2 and len(segs_a) >= 4                                  # its own 2-frame schedule
    # a loop that waits on an empty feed ends when the feed is closed
    feed2 = RequestFeed()
    t = threading.Timer(0.2, feed2.close)
    t.start()
    assert list(serve_stream(model=model, codec=codec, requests=feed2, max_batch=2)) == []
    with pytest.raises(RuntimeError):
        feed2.put(a)


def test_serve_stream_feed_a_bad_request_fails_alone():
    """from a RequestFeed (a server) a request that cannot be admitted -- a prompt as long as the cache, a
    malformed prompt -- gets an "error" event of its own; the loop and the other requests' utterances go on.  From a
    plain list (a batch caller) the same request raises, as before."""
    from fish_speech_amd.serving import RequestFeed, StreamRequest, collect, serve_stream

    model, codec = StubDualAR(max_batch=2, max_seq_len=64), StubCodec()
    good = StreamRequest(prompt=torch.zeros(NCB + 1, 5, dtype=torch.int64), seed=11, rid=1, max_new_tokens=6)
    too_long = StreamRequest(prompt=torch.zeros(NCB + 1, 64, dtype=torch.int64), seed=3, rid=2)
    malformed = StreamRequest(prompt=torch.zeros(NCB, 5, dtype=torch.int64), seed=4, rid=3)
    other = StreamRequest(prompt=torch.zeros(NCB + 1, 7, dtype=torch.int64), seed=5, rid=4, max_new_tokens=6)
    one_dim = StreamRequest(prompt=torch.zeros(5, dtype=torch.int64), seed=6, rid=5)      # .size(1) came first
    not_a_tensor = StreamRequest(prompt=[[0] * 5] * (NCB + 1), seed=7, rid=6)
    feed = RequestFeed()
    for r in (good, too_long, malformed, one_dim, not_a_tensor, other):
        feed.put(r)
    evs = list(serve_stream(model=model, codec=codec, requests=feed, max_batch=2, step_frames=4, return_when_idle=True))
    errs = {e.rid: e.error for e in evs if e.kind == "error"}
    assert set(errs) == {2, 3, 5, 6} and "exceeds max_seq_len" in errs[2] and "prompt must be" in errs[3]
    assert "prompt must be" in errs[5] and "prompt must be" in errs[6]
    assert sorted(e.rid for e in evs if e.kind == "final") == [1, 4] and not model.slots
    got = collect(evs, codec.frame_length)
    assert torch.equal(got[1][1], model.expected_codes(11, 6)) and torch.equal(got[4][1], model.expected_codes(5, 6))
    with pytest.raises(ValueError, match="exceeds max_seq_len"):
        list(serve_stream(model=model, codec=codec, requests=[good, too_long], max_batch=2))
    assert not model.slots


@pytest.mark.timeout(60)
def test_engine_lock_excludes_interleaved_generators_and_may_be_released_by_another_thread():
    """`inference()` holds the model's lock across yields.  (a) Two requests interleaved on ONE thread must
    not both drive slot 0: the second reports an error result (after `lock_timeout`) instead of passing a re-entrant
    lock and corrupting the first one's audio, and the first still yields exactly its own result.  (b) A generator that
    was advanced on one thread may be closed from another (thread-pool iteration of a streaming response): the lock
    is released there and the next request runs."""
    import threading

    model, codec = StubDualAR(max_batch=1), StubCodec()
    model.lock = threading.Lock()
    eng = StreamingTTSEngine(model, codec, precision=None)
    eng.lock_timeout = 0.2
    mk = lambda seed: TTSRequest(text="hello there", streaming=True, max_new_tokens=64, seed=seed, first_chunk_frames=2, chunk_frames=3)  # noqa: E731
    want = list(eng.inference(mk(3)))
    g1 = eng.inference(mk(3))
    first = [next(g1), next(g1)]                       # header + first segment: the lock is held now
    g2 = eng.inference(mk(4))
    res2 = list(g2)                                     # same thread, first request suspended
    assert [r.code for r in res2] == ["error"] and isinstance(res2[0].error, TimeoutError)
    rest = list(g1)
    got = first + rest
    assert [r.code for r in got] == [r.code for r in want]
    assert np.array_equal(got[-1].audio[1], want[-1].audio[1])
    assert model.lock.acquire(blocking=False)
    model.lock.release()
    # (b) advanced here, closed on another thread
    g3 = eng.inference(mk(5))
    assert next(g3).code == "header" and next(g3).code == "segment" and model.lock.locked()
    err = []

    def closer():
        try:
            g3.close()
        except Exception as e:   # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=closer)
    t.start()
    t.join(10)
    assert not err and not model.lock.locked() and not model.slots
    assert list(eng.inference(mk(3)))[-1].code == "final"


@pytest.mark.timeout(60)
def test_batching_engine_serves_concurrent_requests_through_one_loop():
    """BatchingTTSEngine: request threads share one serve_stream loop (continuous batching) instead of taking turns;
    every request's result equals the serial engine's, the protocol is unchanged, several utterances were in flight
    together, an abandoned request frees its slot, errors come back as results, and the loop lets go of the model's
    lock between bursts."""
    import threading

    from fish_speech_amd.engine import BatchingTTSEngine

    codec = StubCodec()
    serial = StreamingTTSEngine(StubDualAR(max_batch=1), codec, precision=None)
    model = StubDualAR(max_batch=4)
    model.lock = threading.Lock()
    peak = [0]
    orig_decode = model.decode

    def decode(slots, n):
        peak[0] = max(peak[0], len(slots))
        time.sleep(0.002)                     # give the other request threads time to arrive while this one is live
        return orig_decode(slots, n)

    model.decode = decode
    eng = BatchingTTSEngine(model, codec, precision=None, max_batch=4, step_frames=4)
    reqs = [TTSRequest(text=f"hello there {i}", streaming=bool(i % 2), max_new_tokens=64, seed=2 + i, first_chunk_frames=2,
                       chunk_frames=3) for i in range(6)]
    want = [list(serial.inference(r)) for r in reqs]
    got = [None] * len(reqs)

    def run(i):
        got[i] = list(eng.inference(reqs[i]))

    ths = [threading.Thread(target=run, args=(i,)) for i in range(len(reqs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(30)
    assert all(g is not None for g in got) and peak[0] >= 2
    for r, g, w in zip(reqs, got, want):
        assert g[-1].code == "final" and np.array_equal(g[-1].audio[1], w[-1].audio[1])
        assert [x.code for x in g][:1] == (["header"] if r.streaming else ["final"])
        if r.streaming:
            assert np.array_equal(np.concatenate([x.audio[1] for x in g[1:-1]]), g[-1].audio[1])
    time.sleep(0.1)
    assert not model.slots and model.lock.acquire(blocking=False)      # idle: nothing held
    model.lock.release()
    # a consumer that walks away after the first segment: its utterance is cancelled and the slot comes back
    gen = eng.inference(TTSRequest(text="a long one", streaming=True, max_new_tokens=64, seed=11, first_chunk_frames=1, chunk_frames=1))
    assert next(gen).code == "header" and next(gen).code == "segment"
    gen.close()
    for _ in range(200):
        if not model.slots:
            break
        time.sleep(0.01)
    assert not model.slots
    # errors are results, and the loop survives them
    res = list(eng.inference(TTSRequest(text="hi", max_new_tokens=1, seed=1)))
    assert [r.code for r in res] == ["error"] and "No audio generated" in str(res[0].error)
    again = list(eng.inference(reqs[0]))
    assert np.array_equal(again[-1].audio[1], want[0][-1].audio[1])
    eng.close()


def test_engine_reports_errors_as_results():
    model, codec = StubDualAR(max_batch=1, max_seq_len=2100), StubCodec()
    eng = StreamingTTSEngine(model, codec, precision=None)
    # prompt longer than max_seq_len - 2048 (text2semantic/inference.py:658-661)
    res = list(eng.inference(TTSRequest(text="x" * 200, max_new_tokens=8, seed=1)))
    assert [r.code for r in res] == ["error"] and "too long" in str(res[0].error)
    with pytest.raises(RuntimeError, match="too long"):
        list(inference_wrapper(TTSRequest(text="x" * 200, max_new_tokens=8, seed=1), eng))
    # an utterance of one frame has no voiced frame at all
    one = StubDualAR(max_batch=1)
    res = list(StreamingTTSEngine(one, codec, precision=None).inference(TTSRequest(text="hi", max_new_tokens=1, seed=1)))
    assert [r.code for r in res] == ["error"] and "No audio generated" in str(res[0].error)
    assert not model.slots and not one.slots


def _wav_bytes(x, sr):
    buf = io.BytesIO()
    with wave.open(buf, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype(np.int16).tobytes())
    return buf.getvalue()


def test_server_batch_helpers_pad_trim_and_cache():
    codec = StubCodec()
    # decode: ragged code matrices, more items than one micro-batch
    feats = [torch.randint(0, 1024, (NCB + 1, t), generator=torch.Generator().manual_seed(t)) for t in (5, 1, 9, 3, 7, 2, 8, 4, 6, 11)]
    outs = server_utils.batch_vqgan_decode(codec, feats)
    assert len(outs) == len(feats) > server_utils.MICRO_BATCH_SIZE
    for f, o in zip(feats, outs):
        assert o.shape == (1, f.shape[-1] * codec.frame_length)
        assert np.array_equal(o, codec.from_indices(f[None])[0].numpy())
    # encode: tensors and wav bytes (one at another sample rate), trimmed to each item's frames
    t = np.linspace(0, 1, 4410, endpoint=False)
    a0 = torch.from_numpy((0.3 * np.sin(2 * np.pi * 50 * t)).astype(np.float32))[None]
    wav_same = _wav_bytes(0.25 * np.ones(1000), 44100)
    wav_half = _wav_bytes(0.5 * np.ones(600), 22050)
    res = server_utils.batch_encode(codec, [a0, wav_same, wav_half])
    assert [r.shape for r in res] == [(NCB + 1, -(-4410 // 16)), (NCB + 1, -(-1000 // 16)), (NCB + 1, -(-1200 // 16))]
    assert int(res[1][0, 3]) == 250 and abs(int(res[2][0, 10]) - 500) <= 2      # resampled to 44.1 kHz
    # LRU keyed by the byte strings
    server_utils._cache.clear()
    n_before = len(codec.calls)
    r1 = server_utils.cached_vqgan_batch_encode(codec, [wav_same])
    r2 = server_utils.cached_vqgan_batch_encode(codec, [wav_same])
    assert r1 is r2 and len(server_utils._cache) == 1 and len(codec.calls) == n_before
    server_utils.cached_vqgan_batch_encode(codec, [wav_half])
    assert len(server_utils._cache) == 2
    server_utils._cache.clear()


def test_inference_result_dataclass_matches_the_reference_fields():
    r = InferenceResult(code="final", audio=(44100, np.zeros(2, np.float32)), error=None)
    assert (r.code, r.error) == ("final", None) and r.audio[0] == 44100


def _bare_model(vocab=300, cbs=16, ncb=3):
    """MiDualAR without a device library: only the pure-host helpers are touched."""
    from fish_speech_amd.dual_ar import MiDualAR

    m = object.__new__(MiDualAR)
    m.config = SimpleNamespace(vocab_size=vocab, codebook_size=cbs, num_codebooks=ncb)
    m._cached_prompt = {}
    return m


def test_prompt_token_range_check_mirrors_embedding_index_errors():
    m = _bare_model()
    ok = torch.tensor([[299, 15, 0, 7], [0, 0, 0, 0]])
    m._check_tokens(ok)
    m._check_tokens(ok[:0])
    for bad in ([[300, 0, 0, 0]], [[-1, 0, 0, 0]], [[5, 16, 0, 0]], [[5, 0, -2, 0]]):
        with pytest.raises(IndexError):
            m._check_tokens(torch.tensor(bad))


def test_reusable_prefix_rules():
    """Longest common prefix, at least one column left to run, and never across the two prefill kernel classes
    (<= 16 rows: decode GEMV, longer: tiled GEMM -- K/V written by one are not bit-identical to the other's)."""
    m = _bare_model()
    g = torch.Generator().manual_seed(0)
    base = torch.randint(0, 16, (4, 60), generator=g)
    assert m._reusable_prefix(0, base) == 0                       # nothing cached
    m._cached_prompt[0] = base[:, :40].clone()
    assert m._reusable_prefix(0, base[:, :50]) == 40              # extension: the whole cached prompt
    assert m._reusable_prefix(0, base[:, :40]) == 39              # same prompt: one column still runs
    assert m._reusable_prefix(0, base[:, :30]) == 29              # shorter prompt
    changed = base[:, :50].clone()
    changed[2, 17] ^= 1
    assert m._reusable_prefix(0, changed) == 17                   # first differing column (any row)
    assert m._reusable_prefix(1, base[:, :50]) == 0               # another slot
    assert m._reusable_prefix(0, base[:, :12]) == 0               # long cached, short new: other kernel class
    m._cached_prompt[0] = base[:, :9].clone()
    assert m._reusable_prefix(0, base[:, :14]) == 9               # both short
    assert m._reusable_prefix(0, base[:, :30]) == 0               # short cached, long new
    assert m._reusable_prefix(0, base[:, :1]) == 0                # a single column always runs


def test_from_indices_ragged_groups_by_length_pads_at_the_end_and_keeps_the_callers_order():
    """MiDAC.from_indices_ragged's host logic on a stand-in codec (the GPU test checks the audio bit for bit): utterances
    are grouped by length, at most `max_group` per call, a group's padding bounded by `pad_waste`, every utterance's
    codes reach the decoder unchanged in front of the padding, results come back in the caller's order, an empty
    utterance yields an empty waveform without a call."""
    from types import SimpleNamespace

    from fish_speech_amd.dac import MiDAC

    calls = []

    class Fake:
        config = SimpleNamespace(n_codebooks=3)
        device = torch.device("cpu")
        frame_length = 4
        module_dtype = torch.float32

        def from_indices(self, batch):
            calls.append(tuple(batch.shape))
            # "audio" = the first codebook's code of every frame, repeated frame_length times: recognisable per utterance
            return batch[:, :1, :].float().repeat_interleave(self.frame_length, dim=2)

    lens = [9, 1, 30, 10, 0, 29, 9, 3]
    codes = [torch.full((4, t), 100 + i, dtype=torch.int64) for i, t in enumerate(lens)]
    out = MiDAC.from_indices_ragged(Fake(), codes, max_group=3, pad_waste=0.35)
    assert len(out) == len(lens)
    for i, t in enumerate(lens):
        assert out[i].shape == (1, 1, 4 * t)
        assert t == 0 or bool((out[i] == 100 + i).all())            # its own codes, none of the padding's zeros
    assert all(b <= 3 for b, _, _ in calls)
    # (30, 29, 10): 21 padded frames <= 0.35 x 69 real ones | (9, 9, 3): 6 <= 0.35 x 21 | a 4th member would exceed max_group | (1)
    assert sorted(calls) == sorted([(3, 4, 30), (3, 4, 9), (1, 4, 1)]), calls
    calls.clear()
    MiDAC.from_indices_ragged(Fake(), codes, max_group=8, pad_waste=0.0)      # no padding allowed: equal lengths only
    assert sorted(calls) == sorted([(1, 4, 30), (1, 4, 29), (1, 4, 10), (2, 4, 9), (1, 4, 3), (1, 4, 1)]), calls
    with pytest.raises(ValueError):
        MiDAC.from_indices_ragged(Fake(), [torch.zeros(5, 7, dtype=torch.int64)])

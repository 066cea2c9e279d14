"""Where streaming time goes (config 5): per-chunk incremental codec decodes vs one offline decode, at the bench size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_codec_state
from fish_speech_amd.dac import DacConfig, MiDAC
from fish_speech_amd.stream import chunk_schedule

dev = torch.device("cuda:0")
cfg = DacConfig()
codec = MiDAC(cfg, device=dev)
codec.load_folded_state(synthetic_codec_state(cfg, dev))
B, T = 8, 215
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.randint(0, 1024, (B, 10, T), generator=g, device=dev, dtype=torch.int64)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
full = timed(lambda: codec.from_indices(codes.clone()))
marks = chunk_schedule(T, 8, 32)
only = os.environ.get("STREAM_ONLY", "")     # "cached": just the stream_id path (for a kernel trace of the chunked decode alone)
for cached in ((True,) if only == "cached" else (False, True)):
    # cached: the quantizer-side state of frames [0, t0) is kept between the calls of a stream.  Each chunk is timed on
    # the state its predecessor left (a repeated call would not continue it, so the
    # whole stream is replayed for every repetition).
    tot, per = 0.0, []
    for rep in range(3):
        sid = codec.new_stream_id() if cached else None
        t0, cur = 0, []
        for t1 in marks:
            c = codes[:, :, :t1].clone()
            torch.cuda.synchronize(); w0 = time.perf_counter()
            codec.from_indices_tail(c, t0, stream_id=sid)
            torch.cuda.synchronize(); cur.append((time.perf_counter() - w0) * 1e3)
            t0 = t1
        per = cur if not per else [min(a, b) for a, b in zip(per, cur)]
    t0 = 0
    print(f"# {'state kept between calls (stream_id)' if cached else 'quantizer side recomputed per call'}")
    for t1, ms in zip(marks, per):
        print(f"frames [{t0:3d},{t1:3d}): {ms:6.2f} ms  ({ms / (t1 - t0):.3f} ms/frame; offline {full / T:.3f})", flush=True)
        t0 = t1
    print(f"offline decode {full:.1f} ms; sum of incremental decodes {sum(per):.1f} ms")

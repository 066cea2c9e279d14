"""Kernel-level view of the FIRST streamed chunk (8 frames, batch 8): `rocprofv3 --kernel-trace --stats -- python
tools/first_chunk_probe.py [frames] [reps]` -- every repetition opens a new stream and decodes frames [0, frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_codec_state
from fish_speech_amd.dac import DacConfig, MiDAC

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
cfg = DacConfig()
codec = MiDAC(cfg, device=dev)
codec.load_folded_state(synthetic_codec_state(cfg, dev))
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.randint(0, 1024, (8, 10, frames), generator=g, device=dev, dtype=torch.int64)
ts = []
for r in range(reps + 2):
    sid = codec.new_stream_id()
    c = codes.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    codec.from_indices_tail(c, 0, stream_id=sid)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    codec.close_stream(sid)
print(f"first chunk of {frames} frames, batch 8: min {min(ts[2:]):.2f} ms, median {sorted(ts[2:])[len(ts[2:]) // 2]:.2f} ms over {reps} streams")

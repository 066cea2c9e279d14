"""One incremental codec decode (frames [T0, T1) of 8 utterances) repeated: run under rocprofv3 --kernel-trace.
CACHED=1: the quantizer-side state of frames [0, T0) is kept (stream_id), as generate_stream does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_codec_state
from fish_speech_amd.dac import DacConfig, MiDAC

dev = torch.device("cuda:0")
cfg = DacConfig()
codec = MiDAC(cfg, device=dev)
codec.load_folded_state(synthetic_codec_state(cfg, dev))
B, T0, T1 = 8, int(os.environ.get("T0", 40)), int(os.environ.get("T1", 72))
cached = os.environ.get("CACHED", "1") == "1"
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.randint(0, 1024, (B, 10, T1), generator=g, device=dev, dtype=torch.int64)
for _ in range(int(os.environ.get("N", 5))):
    if cached:
        sid = codec.new_stream_id()
        codec.from_indices_tail(codes[:, :, :T0].clone(), 0, stream_id=sid)
        torch.cuda.synchronize()
        time.sleep(0.1)          # an idle gap: tools/rocpd_clusters.py splits the trace into bursts there
        c = codes.clone()
        torch.cuda.synchronize()
        time.sleep(0.1)
        codec.from_indices_tail(c, T0, stream_id=sid)
        torch.cuda.synchronize()
        time.sleep(0.1)
    else:
        codec.from_indices_tail(codes.clone(), T0)
torch.cuda.synchronize()

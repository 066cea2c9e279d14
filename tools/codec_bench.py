"""Time MiDAC.from_indices at the benchmark size (B=8, T=215, yaml-sized codec, random weights)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_codec_state
from fish_speech_amd.dac import DacConfig, MiDAC

dev = torch.device("cuda:0")
cfg = DacConfig()
codec = MiDAC(cfg, device=dev)
codec.load_folded_state(synthetic_codec_state(cfg, dev))
B, T = int(os.environ.get("B", 8)), int(os.environ.get("T", 215))
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.randint(0, 1024, (B, 10, T), generator=g, device=dev, dtype=torch.int64)
codec.from_indices(codes.clone()); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
flop = 2 * 727.15e9 * B * T / 215
for planes in [int(v) for v in os.environ.get("PLANES", "2,0,1").split(",")]:
    codec.set_precision(planes)
    codec.from_indices(codes.clone()); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0.record(); out = codec.from_indices(codes.clone()); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"planes={planes} FMI_CONV_MT={os.environ.get('FMI_CONV_MT', '-')} decode B={B} T={T}: {min(ts):.1f} ms  ({flop / min(ts) / 1e9:.1f} TFLOP/s useful)  checksum {float(out.double().abs().sum()):.6f}", flush=True)
codec.set_precision(0)
for dbg in os.environ.get("DBG_MODES", "").split(","):
    if not dbg:
        continue
    os.environ["FMI_CONV_DBG"] = dbg
    ts = []
    for _ in range(3):
        e0.record(); out = codec.from_indices(codes.clone()); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"dbg={dbg} prefetch={os.environ.get('FMI_CONV_PREFETCH','1')} decode B={B} T={T}: {min(ts):.1f} ms  ({flop / min(ts) / 1e9:.1f} TFLOP/s)  checksum {float(out.double().abs().sum()):.6f}", flush=True)
os.environ["FMI_CONV_DBG"] = "0"
if os.environ.get("ENCODE", "1") == "1":
    a = 0.1 * torch.randn(B, 1, 3 * 44100, generator=g, device=dev)
    codec.encode(a); torch.cuda.synchronize()
    e0.record(); c, l = codec.encode(a); e1.record(); e1.synchronize()
    print(f"encode B={B} 3 s: {e0.elapsed_time(e1):.1f} ms  ({2 * 115.8e9 * B / e0.elapsed_time(e1) / 1e9:.1f} TFLOP/s useful)  codes sum {int(c.sum())}")

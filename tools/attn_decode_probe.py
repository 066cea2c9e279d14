"""Decode attention at long contexts: 8 utterances with T-token prompts, 12 decode frames, at the
S2-Pro shape.  Run under `rocprofv3 --kernel-trace --stats` once per T (tools/make_profiles.sh): the summary line of
attn_decode_fused_kernel gives its average duration at context ~T; algorithmic bytes per launch =
B x 2 (K, V) x KVH x D x 2 B x S = 32768 S, which a bare streaming read of that size bounds from below.
usage: python tools/attn_decode_probe.py T"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fish_speech_amd.dual_ar import MiDualAR

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
cfg = bench.s2_pro_config(max_seq_len=T + 64)
model = MiDualAR(cfg, device=dev, im_end_id=cfg.im_end_id)
model.load_state_dict(bench.synthetic_state_on_device(cfg, dev))
model.setup_caches(8, cfg.max_seq_len)
model.set_ignore_eos(True)
bench.PROMPT_T = T
prompts = bench.make_prompts(cfg, 8, 1000)
sp = [model._sampling(0.7, 0.7, 30, 4242 + i, True) for i in range(8)]
model.prefill(list(range(8)), prompts, [40] * 8, sp)
model.decode(list(range(8)), 4)      # graph build + warm-up
model.synchronize()
model.decode(list(range(8)), 12)
model.synchronize()
ms, n = model.last_decode_stats()
print(f"T={T}: decode frame {ms / 12:.3f} ms at context {T + 4}..{T + 16}; attention bytes per launch {32768 * (T + 10) / 1e6:.1f} MB", flush=True)

"""Prefill timing at the S2-Pro shape: 8 prompts of T tokens (bench.py's model), MFMA flash attention vs the VALU
kernel.  usage: python tools/prefill_bench.py [T ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fish_speech_amd.dual_ar import MiDualAR

dev = torch.device("cuda:0")
Ts = [int(a) for a in sys.argv[1:]] or [200, 1024, 2048]
cfg = bench.s2_pro_config(max_seq_len=max(Ts) + 64)
model = MiDualAR(cfg, device=dev, im_end_id=cfg.im_end_id)
model.load_state_dict(bench.synthetic_state_on_device(cfg, dev))
model.setup_caches(8, cfg.max_seq_len)
model.set_ignore_eos(True)
for T in Ts:
    bench.PROMPT_T = T
    prompts = bench.make_prompts(cfg, 8, 1000)
    sp = [model._sampling(0.7, 0.7, 30, 4242 + i, True) for i in range(8)]
    for impl in [int(v) for v in os.environ.get("ATTN_IMPLS", "1,0").split(",")]:
        model.set_attn_impl(impl)
        ts = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.prefill(list(range(8)), prompts, [2] * 8, sp)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            for i in range(8):
                model.release(i)
        print(f"T={T} attn_impl={'mfma' if impl else 'valu'}: prefill of 8 prompts {min(ts) * 1e3:.2f} ms", flush=True)

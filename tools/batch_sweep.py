#!/usr/bin/env python3
"""Decode-frame time and throughput over the batch size (S2-Pro shape, 200-token prompts, 215 frames, no codec):
what the weight-streaming loop gives per GPU when more utterances share one pass over the weights.  BASELINE.json quotes
batch 8; this is evidence for serving design, not a bench line.  usage: python tools/batch_sweep.py [B ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from fish_speech_amd.dual_ar import DualARConfig, MiDualAR, generate_batch_device


def main():
    bs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 16]
    dev = "cuda:0"
    cfg = bench.s2_pro_config()
    model = MiDualAR(cfg, device=dev, im_end_id=cfg.im_end_id)
    model.load_state_dict(bench.synthetic_state_on_device(cfg, dev))
    model.setup_caches(max(bs), cfg.max_seq_len)
    model.set_ignore_eos(True)
    for B in bs:
        prompts = bench.make_prompts(cfg, B, 1000)
        seeds = [4242 + i for i in range(B)]
        run = lambda: generate_batch_device(model=model, prompts=prompts, max_new_tokens=bench.N_FRAMES, seeds=seeds,
                                            temperature=0.7, top_p=0.7, top_k=30)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, _ = model.last_decode_stats()        # the last decode call: N_FRAMES - 1 graph-replayed frames
        audio = B * bench.N_FRAMES * bench.FRAME_LEN / bench.SAMPLE_RATE
        print(f"B={B:2d}: prefill + 214 frames {dt * 1e3:7.1f} ms, decode frame {ms / (bench.N_FRAMES - 1):.3f} ms, "
              f"{audio / dt:6.1f} audio-s/s without the codec", flush=True)


if __name__ == "__main__":
    main()

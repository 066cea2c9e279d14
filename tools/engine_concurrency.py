#!/usr/bin/env python3
"""Concurrent clients against the engine (S2-Pro shape, synthetic weights, 10 s of audio per request):
`StreamingTTSEngine` serves request threads one after the other like upstream's single-worker queue,
`BatchingTTSEngine` lets them share one serve_stream loop.  Prints, for each, aggregate audio-seconds per second and
the first-audio latency seen by the clients.  usage: python tools/engine_concurrency.py [n_clients=16]"""
import os
import re
import statistics
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from fish_speech_amd.dac import DacConfig, MiDAC
from fish_speech_amd.dual_ar import MiDualAR
from fish_speech_amd.engine import BatchingTTSEngine, StreamingTTSEngine, TTSRequest


class ByteTok:
    """bytes + the specials the prompt builder asks for, mapped into the S2 vocabulary's id ranges"""

    def __init__(self, cfg):
        names = ["<|endoftext|>", "<|pad|>", "<|im_start|>", "<|phoneme_start|>", "<|phoneme_end|>", "<|text|>", "<|voice|>",
                 "<|interleave|>", "<|audio_start|>", "<|audio_end|>", "<|audio_pad|>"] + [f"<|speaker:{i}|>" for i in range(16)]
        self.vocab = {t: 1000 + i for i, t in enumerate(names)}
        self.vocab["<|im_end|>"] = cfg.im_end_id
        self.semantic_begin_id, self.semantic_end_id = cfg.semantic_begin_id, cfg.semantic_end_id
        self._pat = re.compile(r"(<\|[a-z_]+(?::\d+)?\|>)")

    def get_token_id(self, t):
        return self.vocab[t]

    def encode(self, text, add_special_tokens=False, **kw):
        out = []
        for piece in self._pat.split(text):
            if piece:
                out.extend([self.vocab[piece]] if piece in self.vocab else list(piece.encode("utf-8")))
        return out

    def decode(self, ids, **kw):
        return bytes(i for i in ids if i < 256).decode("utf-8", "replace")


def run(engine, reqs):
    first, done, lock = {}, {}, threading.Lock()
    t0 = time.perf_counter()

    def client(i):
        torch.cuda.set_device(0)
        n = 0
        for r in engine.inference(reqs[i]):
            if r.code == "segment":
                with lock:
                    first.setdefault(i, time.perf_counter() - t0)
                n += r.audio[1].size
            elif r.code == "error":
                raise r.error
        with lock:
            done[i] = n

    ths = [threading.Thread(target=client, args=(i,)) for i in range(len(reqs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    lat = sorted(first.values())
    return sum(done.values()) / bench.SAMPLE_RATE / wall, wall, lat


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = "cuda:0"
    cfg = bench.s2_pro_config(max_seq_len=4096)
    model = MiDualAR(cfg, device=dev, im_end_id=cfg.im_end_id)
    model.load_state_dict(bench.synthetic_state_on_device(cfg, dev))
    model.tokenizer = ByteTok(cfg)
    model.setup_caches(8, cfg.max_seq_len)
    model.set_ignore_eos(True)
    ccfg = DacConfig()
    codec = MiDAC(ccfg, device=dev)
    codec.load_folded_state(bench.synthetic_codec_state(ccfg, dev))
    text = "<|speaker:0|>" + "The quick brown fox jumps over the lazy dog, again and again. " * 2
    reqs = [TTSRequest(text=text, streaming=True, max_new_tokens=bench.N_FRAMES + 1, seed=100 + i, chunk_length=400)
            for i in range(n)]
    for name, eng in (("StreamingTTSEngine (one request at a time)", StreamingTTSEngine(model, codec)),
                      ("BatchingTTSEngine  (shared serve_stream loop, 8 slots)", BatchingTTSEngine(model, codec, max_batch=8))):
        run(eng, reqs[:2])                                            # warm-up: graphs, codec buffers
        rate, wall, lat = run(eng, reqs)
        print(f"{name}: {n} clients x 10 s of audio in {wall:.2f} s = {rate:.1f} audio-s/s; first audio p50 "
              f"{statistics.median(lat) * 1e3:.0f} ms, p90 {lat[int(0.9 * (len(lat) - 1))] * 1e3:.0f} ms, last "
              f"{lat[-1] * 1e3:.0f} ms", flush=True)
        if hasattr(eng, "close"):
            eng.close()


if __name__ == "__main__":
    main()

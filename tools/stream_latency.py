"""BASELINE.json config 5: streaming chunked decode at batch 8 on one MI355X -- p50 first-audio latency.

S2-Pro-shaped random weights, eight 200-token prompts, 10 s of audio each (215 frames), sampled
(temperature 0.7 / top-p 0.7 / top-k 30).  `first audio` = wall time from the call of generate_stream
until the first audio chunk (first_chunk_frames frames of every utterance) is complete on the device:
prefill + first_chunk_frames decode frames + one incremental codec decode.  Also reports the cost of
streaming the whole utterance against the offline path (bench.py's step).

usage: python tools/stream_latency.py [--first 8] [--chunk 32] [--runs 9]"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--runs", type=int, default=9)
    ap.add_argument("--growth", type=float, default=1.0, help="every later chunk is this much longer (chunk_schedule)")
    ap.add_argument("--timing", action="store_true", help="host wall time per phase and chunk of the last run")
    args = ap.parse_args()
    device = torch.device("cuda:0")
    from fish_speech_amd.dac import DacConfig, MiDAC
    from fish_speech_amd.dual_ar import MiDualAR
    from fish_speech_amd.stream import generate_stream

    cfg = bench.s2_pro_config()
    model = MiDualAR(cfg, device=device, im_end_id=cfg.im_end_id)
    model.load_state_dict(bench.synthetic_state_on_device(cfg, device))
    model.setup_caches(bench.BATCH, bench.PROMPT_T + bench.N_FRAMES + 8)
    model.set_ignore_eos(True)
    codec = MiDAC(DacConfig(), device=device)
    codec.load_folded_state(bench.synthetic_codec_state(DacConfig(), device))
    prompts = bench.make_prompts(cfg, bench.BATCH, 1000)
    seeds = [4242 + i for i in range(bench.BATCH)]
    n_new = bench.N_FRAMES + 1   # the last generated frame is never voiced (inference.py:708)

    timing = []

    def one_run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first, n_samples, marks = None, 0, []
        timing.clear()
        for ch in generate_stream(model=model, codec=codec, prompts=prompts, max_new_tokens=n_new,
                                  first_chunk_frames=args.first, chunk_frames=args.chunk, seeds=seeds,
                                  temperature=0.7, top_p=0.7, top_k=30, timing=timing if args.timing else None,
                                  chunk_growth=args.growth):
            ts = time.perf_counter()
            torch.cuda.synchronize()
            if timing:
                timing[-1]["consumer_sync"] = time.perf_counter() - ts
            if first is None:
                first = time.perf_counter() - t0
            n_samples += ch.audio.shape[-1]
            marks.append(ch.t1)
        return first, time.perf_counter() - t0, n_samples, marks

    one_run()  # warm-up: graph capture, workspace allocation
    firsts, totals = [], []
    for _ in range(args.runs):
        f, t, n_samples, marks = one_run()
        firsts.append(f * 1e3)
        totals.append(t)
    # offline reference point: the same work without streaming
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_step(model, codec, prompts, seeds, device)
    torch.cuda.synchronize()
    offline = time.perf_counter() - t0
    audio_s = bench.BATCH * n_samples / bench.SAMPLE_RATE
    print(f"chunks end at frames {marks}")
    for t in timing:
        print("  chunk to frame %3d: " % t["t1"] + "  ".join(f"{k} {v * 1e3:7.2f} ms" for k, v in t.items() if k != "t1"))
    print(f"first audio ({args.first} frames = {args.first * bench.FRAME_LEN / bench.SAMPLE_RATE * 1e3:.0f} ms of audio "
          f"x {bench.BATCH} utterances): p50 {statistics.median(firsts):.1f} ms  min {min(firsts):.1f}  max {max(firsts):.1f}  "
          f"({args.runs} runs)")
    print(f"whole stream: p50 {statistics.median(totals) * 1e3:.0f} ms for {audio_s:.1f} audio-s "
          f"-> {audio_s / statistics.median(totals):.1f} audio-s/s   (offline step {offline * 1e3:.0f} ms)")


if __name__ == "__main__":
    main()

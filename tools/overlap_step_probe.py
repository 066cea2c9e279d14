"""VERDICT r04 #2, measured before anything is kept: the benchmark step with the MFMA-bound codec decode of batch i - 1
running BESIDE the HBM-bound frame loop of batch i (bench.run_steps_overlapped) against the serial step, for several
ways of sharing the chip between the two queues: plain streams, dispatch priorities (frame loop high / codec low), and
CU masks that confine the codec to part of the chip (hipExtStreamCreateWithCUMask).  Prints one line per mode:
ms per step, the decode frame average measured with HIP events on the model's stream, audio-s/s.

    python tools/overlap_step_probe.py [steps] > profiles/r05_overlap_step.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def mask_words(bits):
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    return words


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = bench.s2_pro_config()
    model, codec, _, _ = bench.construct(cfg, dev, 0)
    prompts = bench.make_prompts(cfg, bench.BATCH, 1000)
    seeds = [4242 + i for i in range(bench.BATCH)]
    audio = bench.BATCH * bench.N_FRAMES * bench.FRAME_LEN / bench.SAMPLE_RATE
    side = torch.cuda.Stream(device=dev)

    def serial(tag):
        bench.run_step(model, codec, prompts, seeds, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fms, cms = [], []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(steps):
            codes, _ = bench.run_step(model, None, prompts, seeds, dev)
            fms.append(model.last_decode_stats()[0] / (bench.N_FRAMES - 1))
            e0.record()
            codec.from_indices(codes)
            e1.record()
            e1.synchronize()
            cms.append(e0.elapsed_time(e1))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{tag:58s} step {dt * 1e3:8.2f} ms  frame {sum(fms) / len(fms):.4f} ms  codec {sum(cms) / len(cms):6.2f} ms  "
              f"{audio / dt:6.2f} audio-s/s", flush=True)
        return codes

    def overlapped(tag):
        codec.set_async(True)
        try:
            bench.run_steps_overlapped(model, codec, prompts, seeds, dev, 2, side)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fms, wav = bench.run_steps_overlapped(model, codec, prompts, seeds, dev, steps, side)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        finally:
            codec.set_async(False)
        fr = sorted(fms)
        print(f"{tag:58s} step {dt * 1e3:8.2f} ms  frame {sum(fms) / len(fms):.4f} ms (first {fms[0]:.4f}, others "
              f"{sum(fms[1:]) / max(1, len(fms) - 1):.4f})  {audio / dt:6.2f} audio-s/s", flush=True)
        return wav

    modes = sys.argv[2] if len(sys.argv) > 2 else "plain,prio,mask,floor"
    if modes == "only-serial":      # (for a kernel trace of one regime alone: tools/r05_overlap_trace.sh)
        serial("serial (bench.py r04)")
        return
    if modes == "only-overlapped":
        overlapped("overlapped, plain streams")
        return
    codes = serial("serial (bench.py r04)")
    want = codec.from_indices(codes.clone())
    if "plain" in modes:
        wav = overlapped("overlapped, plain streams")
        assert torch.equal(wav, want), "the overlapped step's waveform differs from the serial step's"
    if "prio" in modes:
        model.set_stream_priority(-1)
        codec.set_stream_options(priority=1)
        overlapped("overlapped, frame loop high / codec low priority")
        serial("serial, same priorities (control)")
        model.set_stream_priority(0)
        codec.set_stream_options(priority=0)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    if "mask" in modes:
        for name, bits in (("first 16 CUs", range(16)), ("first 32 CUs", range(32)), ("first 64 CUs", range(64)),
                           ("every 8th CU (32)", range(0, n_cu, 8)), ("every 4th CU (64)", range(0, n_cu, 4)),
                           ("every 2nd CU (128)", range(0, n_cu, 2))):
            codec.set_stream_options(cu_mask=mask_words(bits))
            wav = overlapped(f"overlapped, codec confined to {name}")
            assert torch.equal(wav, want)
        # the codec alone under a mask: how much longer does the decode take on part of the chip
        for name, bits in (("first 32 CUs", range(32)), ("every 4th CU (64)", range(0, n_cu, 4))):
            codec.set_stream_options(cu_mask=mask_words(bits))
            print(f"codec decode alone, confined to {name}: {codec_alone(codec, codes):.2f} ms", flush=True)
        codec.set_stream_options(priority=0)
    if "floor" in modes:
        # one conv work-group per CU (LDS floor > 80 KiB): the frame loop's GEMV work-groups can be co-resident
        for kib in (56, 84, 120):
            codec.set_background(kib * 1024)
            print(f"codec decode alone, conv LDS floor {kib} KiB: {codec_alone(codec, codes):.2f} ms", flush=True)
            wav = overlapped(f"overlapped, conv LDS floor {kib} KiB")
            assert torch.equal(wav, want)
        codec.set_background(84 * 1024)
        model.set_stream_priority(-1)
        codec.set_stream_options(priority=1)
        overlapped("overlapped, conv LDS floor 84 KiB + loop high / codec low")
        model.set_stream_priority(0)
        codec.set_stream_options(priority=0)
        codec.set_background(0)
    serial("serial again (default streams)")


def codec_alone(codec, codes):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    codec.from_indices(codes.clone())
    e0.record()
    codec.from_indices(codes.clone())
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


if __name__ == "__main__":
    main()

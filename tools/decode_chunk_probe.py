"""What does cutting the frame loop into several fmi_dualar_decode calls cost?  192 frames of the bench model as one
call and as calls of 96 / 32 / 8 / 1 frames, with and without a poll (host sync) after each -- no codec involved.
Round-2 finding: with the caller's stream made to wait on every call's completion event (hipStreamWaitEvent on the
null stream, pending for the length of the call) a frame cost 4.81 ms in one 192-frame call and 5.10 ms in calls of
32; without that wait 4.73 / 4.74 ms (profiles/r02_decode_chunk_probe.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

device = torch.device("cuda:0")
from fish_speech_amd.dual_ar import MiDualAR
cfg = bench.s2_pro_config()
model = MiDualAR(cfg, device=device, im_end_id=cfg.im_end_id)
model.load_state_dict(bench.synthetic_state_on_device(cfg, device))
model.setup_caches(bench.BATCH, bench.PROMPT_T + 4 * 200 + 8)
model.set_ignore_eos(True)
prompts = bench.make_prompts(cfg, bench.BATCH, 1000)
slots = list(range(bench.BATCH))
samp = [model._sampling(0.7, 0.7, 30, 4242 + i, True) for i in slots]
for chunk, poll in ((192, False), (96, False), (32, False), (32, True), (8, True), (1, True)):
    model.prefill(slots, prompts, [200] * bench.BATCH, samp)
    model.decode(slots, 4); model.poll_done(slots)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    done = 0
    while done < 192:
        model.decode(slots, chunk)
        if poll:
            model.poll_done(slots)
        done += chunk
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, nl = model.last_decode_stats()
    print(f"192 frames in calls of {chunk:3d}{' + poll' if poll else '       '}: {dt * 1e3:8.2f} ms  ({dt * 1e3 / 192:.3f} ms/frame; last call's event time {ms / chunk:.3f} ms/frame)", flush=True)
    for s in slots:
        model.release(s)

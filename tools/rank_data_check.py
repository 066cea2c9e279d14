import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cfg = bench.s2_pro_config()
model, codec, _, _ = bench.construct(cfg, dev, 0)
for r in range(8):
    prompts = bench.make_prompts(cfg, bench.BATCH, 1000 + r * bench.BATCH)
    seeds = [4242 + r * bench.BATCH + i for i in range(bench.BATCH)]
    codes, wav = bench.run_step(model, codec, prompts, seeds, dev)
    torch.cuda.synchronize()
    print("rank data", r, "ok", int(codes.sum()), float(wav.abs().mean()), flush=True)
print("ALL_OK")

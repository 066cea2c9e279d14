#!/usr/bin/env python3
"""Find (weights recipe, prompt seed, uniform seed) for the S2-width SAMPLED fixture whose run is robust and leaves the
top-1 candidate (VERDICT r03 weak #1a) -- on the GPU, where an attempt costs a second instead of the minutes the CPU
oracle needs at 4.56 B parameters (oracle/gen_golden_s2.py tried 167 uniform seeds on the CPU, none passed).
The HIP path is only the SEARCH ENGINE here: the winner is then re-run by the unmodified reference on the authoring
container's CPU (S2_HOT / S2_HOT_EVERY / S2_PSEED0 / S2_USEED0 python -m oracle.gen_golden_s2 s2_sampled2), which
re-checks every decision's robustness on ITS OWN logits before writing tests/golden/dualar_s2_sampled2.npz.
usage (GPU box): python tools/search_s2_sampled_gpu.py [hot "1.0,0.99,0.98"] [hot_every] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fish_speech_amd.dual_ar import MiDualAR
from oracle import dual_ar as O

DEV = "cuda:0"
NOISE, TRIALS = 2, 24


def main():
    hot = tuple(float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1.0,0.985,0.97").split(","))
    hot_every = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    budget_s = float(os.environ.get("SEARCH_S", "150"))
    cfg = O.s2_pro_shaped_config(max_seq_len=512)
    skw = dict(seed=2, emb_gain=1.5, slow_gain=200.0, fast_gain=1.0, fast_emb_gain=30.0, hot=hot, hot_every=hot_every, pair_cycles=True)
    model = MiDualAR(cfg, device=DEV, im_end_id=cfg.im_end_id)
    model.load_state_dict(O.make_peaky_state_hash(cfg, device=DEV, **skw))
    model.setup_caches(1, 512)
    model.set_trace(True)
    ids = model._table(1, torch.int32).view(-1).long().cpu()
    dt = torch.bfloat16
    temp, tp, top_k = torch.tensor(0.7, dtype=dt), torch.tensor(0.9, dtype=dt), 30
    hi_t, hi_p = torch.tensor(O.RAS_HIGH_TEMP, dtype=dt), torch.tensor(O.RAS_HIGH_TOP_P, dtype=dt)
    bias_live = O.semantic_logit_bias(cfg, dt)[0, 0][ids]
    ncb1 = cfg.num_codebooks + 1
    t_start = time.time()
    best = (0, None)
    for pseed in range(1, 6):
        prompt = O.make_prompt(cfg, 200, seed=pseed, n_semantic=60)
        T = prompt.shape[1]
        for useed in range(1, 400):
            if time.time() - t_start > budget_s:
                print("budget used; best robust prefix", best, flush=True)
                return
            gen = torch.Generator().manual_seed(1)
            window = torch.zeros(ncb1, O.RAS_WIN_SIZE, dtype=torch.int32)
            last = None
            nt = ras = 0
            ok_frames = 0
            for f in range(frames):
                if f == 0:
                    x, pos0, prev = prompt.t().int().contiguous(), 0, None
                else:
                    x, pos0, prev = last.view(1, ncb1).int().contiguous(), T + f - 1, window.clone()
                sp = model._sampling(0.7, 0.9, top_k, useed, prev is not None)
                out = model.step(x.to(DEV), pos0, sp, prev.to(DEV) if prev is not None else None, f).cpu().long().view(-1)
                live = model.debug_taps(1)[0][0].cpu()
                fast = model.fast_trace(1)[0].cpu()[1:]
                u = lambda d, n: (torch.from_numpy(O.fmi_uniform_u8(useed, 0, f, d, n).astype("float32")) / 256.0).to(dt)   # noqa: E731
                u_n, u_h = u(0, cfg.vocab_size)[ids], u(1, cfg.vocab_size)[ids]
                biased = live + bias_live
                w0 = window[0].clone() if f > 0 else None

                def dec(lg, u_n=u_n, u_h=u_h, w0=w0):
                    a = int(ids[int(O.draw(O.logits_to_probs(lg, temp, tp, top_k), u_n))])
                    if w0 is not None and bool((w0 == a).any()) and cfg.semantic_begin_id <= a <= cfg.semantic_end_id:
                        a = int(ids[int(O.draw(O.logits_to_probs(lg, hi_t, hi_p, top_k), u_h))])
                    return a

                tok = dec(biased)
                if tok != int(out[0]):
                    print(f"  !! frame {f}: the oracle's sampler on the HIP logits gives {tok}, the HIP sampler {int(out[0])}", flush=True)
                    break
                first = int(ids[int(O.draw(O.logits_to_probs(biased, temp, tp, top_k), u_n))])
                ras += int(w0 is not None and bool((w0 == first).any()))
                nt += int(tok != int(ids[int(biased.float().argmax())]))
                good = tok != 0 and O.decision_noise_margin(dec, biased, NOISE, TRIALS, gen)
                for cb in range(1, cfg.num_codebooks):
                    if not good:
                        break
                    u_c = u(1 + cb, cfg.codebook_size)
                    decf = lambda l, u_c=u_c: int(O.draw(O.logits_to_probs(l, temp, tp, top_k), u_c))   # noqa: E731
                    if decf(fast[cb - 1]) != int(out[1 + cb]):
                        print(f"  !! frame {f} cb {cb}: sampler mismatch", flush=True)
                        good = False
                    good = good and O.decision_noise_margin(decf, fast[cb - 1], NOISE, TRIALS, gen)
                if not good:
                    break
                ok_frames += 1
                if f > 0:
                    window = window.roll(-1, dims=1)
                    window[:, -1] = out.int()
                last = out
            if ok_frames > best[0]:
                best = (ok_frames, (pseed, useed, nt, ras))
            done = ok_frames == frames
            print(f"hot {hot} every {hot_every} pseed {pseed} useed {useed}: robust frames {ok_frames}/{frames}, non-top-1 {nt}, RAS {ras}"
                  f"{'  <== CANDIDATE' if done and nt >= 3 and ras >= 8 else ''}", flush=True)
            if done and nt >= 3 and ras >= 8:
                print(f"FOUND S2_HOT={','.join(str(h) for h in hot)} S2_HOT_EVERY={hot_every} S2_PSEED0={pseed} S2_USEED0={useed}", flush=True)
                return


if __name__ == "__main__":
    main()

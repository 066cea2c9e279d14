"""Debug helper (GPU): prints where the HIP Dual-AR path and the oracle differ, with margins."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dual_ar as O
from tests.helpers import load_dualar_case, bf16_close
from tests.test_dualar_gpu import _linear, _linear_oracle, _make_model, DEV
from fish_speech_amd import _lib
from fish_speech_amd.dual_ar import decode_one_token, generate

lib = _lib.load()
# 1. tiled M=300
for norm, epi in ((False, 1), (True, 0), (True, 2)):
    M, path = 300, 2
    g = torch.Generator().manual_seed(M * 7 + epi * 3 + int(norm))
    N, K = (192, 256) if epi != 2 else (2 * 160, 128)
    x = torch.randn(M, K, generator=g).bfloat16(); w = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16() if norm else None
    res = torch.randn(M, N, generator=g).bfloat16() if epi == 1 else None
    got = _linear(lib, x, w, nw, res, M, N, K, epi, path); want = _linear_oracle(x, w, nw, res, epi)
    d = (got.float() - want.float()).abs()
    idx = torch.nonzero(d > 2 * 2**-8 * torch.maximum(got.float().abs(), want.float().abs()) + 1e-3)
    print('tiled', norm, epi, 'maxerr', d.max().item(), 'nbad', len(idx), 'rms want', want.float().pow(2).mean().sqrt().item())
    for i in idx[:5]:
        print('   at', i.tolist(), got[i[0], i[1]].item(), want[i[0], i[1]].item())

# 2. teacher forced + free-run
for case in ('tiny', 'mid'):
    cfg, state, z = load_dualar_case(case)
    prompt = torch.from_numpy(z['prompt']); T = prompt.shape[1]; ncb1 = cfg.num_codebooks + 1
    for mode, tk in (('greedy', 1), ('sampled', 30)):
        orc = O.DualAROracle(cfg, state); orc.trace = {}
        seq = O.generate(orc, prompt, int(z['max_new']), 0.7, 0.7, tk, uniform_fn=O.FmiUniform(1234, 0))
        assert np.array_equal(seq.numpy(), z[mode])
        model = _make_model(cfg, state); model.set_trace(True)
        window = torch.zeros(ncb1, 10, dtype=torch.int32)
        worst_s = worst_h = worst_f = 0.0; nmis = 0
        from fish_speech_amd.dual_ar import SamplingC
        for f in range(seq.shape[1] - T):
            if f == 0:
                x, pos0, prev = prompt.t().int().contiguous(), 0, None
            else:
                x, pos0, prev = seq[:, T + f - 1].view(1, ncb1).int().contiguous(), T + f - 1, window.clone()
            sp = model._sampling(0.7, 0.7, tk, 1234, prev is not None)
            out = model.step(x.to(DEV), pos0, sp, prev.to(DEV) if prev is not None else None, f).cpu()
            logits, ids, hidden, _ = model.debug_taps(1)
            wl = orc.trace['slow_logits'][f][ids.long().cpu()]
            es = (logits[0].float().cpu() - wl.float()).abs().max().item(); worst_s = max(worst_s, es)
            eh = (hidden[0].float().cpu() - orc.trace['hidden'][f].float()).abs().max().item(); worst_h = max(worst_h, eh)
            tr = model.fast_trace(1)[0].cpu()
            if f > 0:
                window = window.roll(-1, dims=1); window[:, -1] = seq[:, T + f].int()
            want = seq[:, T + f]
            if not torch.equal(out.long(), want):
                nmis += 1
                r = int((out.long() != want).nonzero()[0])
                if r == 0:
                    lgo = wl.float(); top = torch.topk(lgo, 3)
                    print(f'  {case}/{mode} frame {f}: slow token {int(out[0])} vs {int(want[0])}; oracle top3 {top.values.tolist()} slow err {es}')
                else:
                    cb = r - 1
                    if cb >= 1:
                        wlf = orc.trace['fast_logits'][f][cb - 1].float(); top = torch.topk(wlf, 3)
                        ef = (tr[cb].float() - wlf).abs().max().item()
                        print(f'  {case}/{mode} frame {f}: cb{cb} {int(out[r])} vs {int(want[r])}; oracle top3 {top.values.tolist()} idx {top.indices.tolist()} fast err {ef}; gpu top {torch.topk(tr[cb].float(),3)}')
                    else:
                        print(f'  {case}/{mode} frame {f}: cb0 {int(out[r])} vs {int(want[r])}')
            # fast logits err only when tokens up to cb agree
            for cb in range(1, cfg.num_codebooks):
                if torch.equal(out[:1 + cb].long(), want[:1 + cb]):
                    ef = (tr[cb].float() - orc.trace['fast_logits'][f][cb - 1].float()).abs().max().item(); worst_f = max(worst_f, ef)
        print(case, mode, 'teacher-forced: frames', seq.shape[1] - T, 'mismatched frames', nmis, 'max err slow', worst_s, 'hidden', worst_h, 'fast', worst_f,
              'logit scale', float(orc.trace['slow_logits'][0].float().abs().max()))

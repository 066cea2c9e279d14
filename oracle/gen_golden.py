"""Generate the committed golden fixtures by running the UNMODIFIED reference (authoring container
only: needs /root/reference).  `python -m oracle.gen_golden`.

Dual-AR fixtures: seeded synthetic weights (oracle.dual_ar.make_synthetic_state, re-creatable from
the seed anywhere torch CPU is the same build) -> the reference's own ``generate`` output tokens,
greedy (top_k=1) and sampled, both with the uniforms of the HIP sampler's generator patched in for
``torch.rand_like`` (even top_k=1 depends on the uniforms upstream: a draw of exactly 0 makes the
reference's exponential race return token 0, which happens once per 256 draws in bf16).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import dual_ar as O
from .refload import add_reference_to_path, build_reference_dual_ar

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

DUALAR_CASES = {
    # name: (config kwargs, state seed, head_gain, prompt T, n_semantic, prompt seed, max_new)
    "tiny": (dict(), 1, 8.0, 24, 8, 5, 24),
    "mid": (dict(vocab_size=2048, n_layer=3, n_head=4, n_local_heads=1, head_dim=128, dim=256,
                 intermediate_size=512, codebook_size=256, num_codebooks=10, semantic_begin_id=1500,
                 semantic_end_id=1755, im_end_id=1400, n_fast_layer=2, max_seq_len=512), 2, 10.0, 40, 12, 9, 20),
}


def _ref_generate(cfg, state, prompt, max_new, top_k, uniform=None):
    add_reference_to_path()
    from fish_speech.models.text2semantic import inference as RI

    ref = build_reference_dual_ar(cfg, state)
    orig_rand, orig_dec = torch.rand_like, RI.decode_one_token_ar
    calls = {"n": 0}

    def dec(*a, **k):
        if uniform is not None and calls["n"] > 0:
            uniform.next_frame()
        calls["n"] += 1
        return orig_dec(*a, **k)

    # The reference's torch.sort(descending=True) is UNSTABLE for more than 16 elements, so the order of
    # exactly-equal bf16 logits -- hence which of them survive top-k / top-p -- is unspecified upstream.
    # The fixture pins that single degree of freedom to "ties in ascending index order" (a valid
    # outcome of the reference algorithm); nothing else of the reference is touched.
    orig_sort = torch.sort

    def stable_sort(x, *a, **k):
        k["stable"] = True
        return orig_sort(x, *a, **k)

    try:
        torch.sort = stable_sort
        RI.decode_one_token_ar = dec  # generate() looks the prefill step up as a module global
        if uniform is not None:
            torch.rand_like = lambda t, **kw: uniform(t.shape[-1], t.dtype)
        y = RI.generate(model=ref, prompt=prompt, max_new_tokens=max_new, audio_masks=None, audio_parts=None,
                        decode_one_token=dec, temperature=0.7, top_p=0.7, top_k=top_k)
    finally:
        torch.rand_like = orig_rand
        torch.sort = orig_sort
        RI.decode_one_token_ar = orig_dec
    return y.long()


def gen_dualar():
    for name, (kw, sseed, gain, T, nsem, pseed, max_new) in DUALAR_CASES.items():
        cfg = O.DualARConfig(**kw)
        prompt = O.make_prompt(cfg, T, seed=pseed, n_semantic=nsem)
        state = O.make_synthetic_state(cfg, seed=sseed, head_gain=gain)
        greedy = _ref_generate(cfg, state, prompt, max_new, 1, O.FmiUniform(seed=1234, stream=0))
        sampled = _ref_generate(cfg, state, prompt, max_new, 30, O.FmiUniform(seed=1234, stream=0))
        np.savez(os.path.join(OUT, f"dualar_{name}.npz"), prompt=prompt.numpy(), greedy=greedy.numpy(),
                 sampled=sampled.numpy(), state_seed=sseed, head_gain=gain, max_new=max_new,
                 uniform_seed=1234, cfg_keys=np.array(list(kw.keys())), cfg_vals=np.array([float(v) for v in kw.values()]))
        print(name, "greedy", tuple(greedy.shape), "sampled", tuple(sampled.shape))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["dualar", "dac"]
    if "dualar" in which:
        gen_dualar()
    if "dac" in which:
        try:
            from .gen_golden_dac import gen_dac
        except ImportError:
            gen_dac = None
        if gen_dac:
            gen_dac()

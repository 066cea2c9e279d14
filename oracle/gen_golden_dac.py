"""Codec golden fixtures from the UNMODIFIED reference DAC (fish_speech/models/dac), imported with the
restated third-party stubs (oracle/stubs/dac, audiotools).  Authoring container only.

The fixture also asserts, at generation time, that oracle/dac.py reproduces the reference bit for bit
on this machine (fp32, same torch build) -- the live pin; tests/ then check the oracle and the HIP
path against the stored arrays anywhere."""
from __future__ import annotations

import math
import os
import time
from functools import partial

import numpy as np
import torch

from . import dac as D
from .refload import add_reference_to_path

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def build_reference_dac(cfg: D.DacConfig, state):
    """modded_dac_vq.yaml, instantiated by hand (hydra/omegaconf are not in this image)."""
    add_reference_to_path()
    from fish_speech.models.dac.modded_dac import DAC, ModelArgs, WindowLimitedTransformer
    from fish_speech.models.dac.rvq import DownsampleResidualVectorQuantize

    L = cfg.latent_dim
    tgc = partial(ModelArgs, block_size=8192, n_local_heads=-1, head_dim=cfg.head_dim, rope_base=cfg.rope_base,
                  norm_eps=cfg.norm_eps, dropout_rate=0.1, attn_dropout_rate=0.1, channels_first=True)
    tgc.window_size = cfg.enc_tf_window  # read via getattr(..., "window_size", 512) at modded_dac.py:641

    def tfm():
        return WindowLimitedTransformer(
            causal=True, window_size=cfg.tf_window, input_dim=L,
            config=ModelArgs(block_size=2048, n_layer=cfg.tf_layers, n_head=L // cfg.head_dim, dim=L,
                             intermediate_size=L * cfg.tf_ffn_mult, n_local_heads=-1, head_dim=cfg.head_dim,
                             rope_base=cfg.rope_base, norm_eps=cfg.norm_eps, dropout_rate=0.1,
                             attn_dropout_rate=0.1, channels_first=True))

    q = DownsampleResidualVectorQuantize(
        input_dim=L, n_codebooks=cfg.n_codebooks, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
        quantizer_dropout=0.5, downsample_factor=list(cfg.downsample), post_module=tfm(), pre_module=tfm(),
        semantic_codebook_size=cfg.semantic_codebook_size)
    n_enc = len(cfg.encoder_rates)
    model = DAC(sample_rate=cfg.sample_rate, encoder_dim=cfg.encoder_dim, encoder_rates=list(cfg.encoder_rates),
                decoder_dim=cfg.decoder_dim, decoder_rates=list(cfg.decoder_rates),
                encoder_transformer_layers=[0] * (n_enc - 1) + [cfg.enc_tf_layers],
                decoder_transformer_layers=[4] + [0] * (len(cfg.decoder_rates) - 1),
                transformer_general_config=tgc, quantizer=q)
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
    assert not unexpected, unexpected[:5]
    assert all(("causal_mask" in m or "freqs_cis" in m) for m in missing), missing[:5]
    return model.eval()


def gen_dac():
    cfg = D.small_config()
    state = D.make_synthetic_state(cfg, seed=11)
    t0 = time.time()
    ref = build_reference_dac(cfg, state)
    print(f"reference DAC built in {time.time() - t0:.1f}s; params {sum(p.numel() for p in ref.parameters())/1e6:.2f}M")
    # the oracle's key table must be exactly the checkpoint's
    ref_keys = {k for k in ref.state_dict().keys() if "causal_mask" not in k and "freqs_cis" not in k}
    assert ref_keys == set(state.keys()), (sorted(ref_keys - set(state))[:5], sorted(set(state) - ref_keys)[:5])
    orc = D.DacOracle(cfg, state)
    g = torch.Generator().manual_seed(3)
    n = 3 * cfg.frame_length - 700  # ragged: exercises the right padding of encode
    t = torch.arange(n) / cfg.sample_rate
    audio = (0.3 * torch.sin(2 * math.pi * 220 * t) + 0.05 * torch.randn(n, generator=g)).view(1, 1, n)
    with torch.no_grad():
        codes_ref, lens_ref = ref.encode(audio.clone(), torch.tensor([n]))
        dec_ref = ref.from_indices(codes_ref.clone())
        rnd = D.make_codes(cfg, 2, 5, seed=4)
        rnd[0, 0, 0] = cfg.semantic_codebook_size + 7      # exercises the in-place clamp (rvq.py:354-359)
        rnd[1, 2, 3] = cfg.codebook_size + 100
        rnd_in = rnd.clone()
        z_ref = ref.quantizer.decode(rnd.clone())
        dec_rnd_ref = ref.from_indices(rnd)
        codes_orc, lens_orc = orc.encode(audio.clone(), torch.tensor([n]))
        dec_orc = orc.from_indices(codes_ref.clone())
        z_orc = orc.dequantize(rnd_in.clone())
        dec_rnd_orc = orc.from_indices(rnd_in.clone())
    assert torch.equal(codes_ref, codes_orc) and torch.equal(lens_ref, lens_orc), "oracle encode != reference"
    for a, b, what in ((dec_ref, dec_orc, "decode"), (z_ref, z_orc, "dequantize"), (dec_rnd_ref, dec_rnd_orc, "decode rnd")):
        err = float((a - b).abs().max())
        print(what, "max |oracle - reference| =", err)
        assert err <= 1e-6, what
    np.savez_compressed(os.path.join(OUT, "dac_small.npz"), audio=audio.numpy(), codes=codes_ref.numpy(),
                        lens=lens_ref.numpy(), decoded=dec_ref.numpy(), rnd_codes=rnd_in.numpy(),
                        rnd_codes_clamped=rnd.numpy(), rnd_z=z_ref.numpy(), rnd_decoded=dec_rnd_ref.numpy(),
                        state_seed=11)
    print("dac_small: audio", tuple(audio.shape), "-> codes", tuple(codes_ref.shape), "-> audio", tuple(dec_ref.shape))


if __name__ == "__main__":
    gen_dac()

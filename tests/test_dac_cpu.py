"""CPU suite for the codec oracle: against the fixtures the unmodified reference DAC produced
(tests/golden/dac_small.npz, oracle/gen_golden_dac.py) and structural properties of the path."""
import os

import numpy as np
import pytest
import torch

from oracle import dac as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dac_small.npz")


@pytest.fixture(scope="module")
def case():
    z = np.load(GOLD)
    cfg = D.small_config()
    return cfg, D.make_synthetic_state(cfg, seed=int(z["state_seed"])), z


def test_oracle_encode_matches_reference_codes(case):
    cfg, state, z = case
    codes, lens = D.DacOracle(cfg, state).encode(torch.from_numpy(z["audio"]), torch.tensor([z["audio"].shape[-1]]))
    assert np.array_equal(codes.numpy(), z["codes"]) and np.array_equal(lens.numpy(), z["lens"])
    assert codes.dtype == torch.int64 and codes.shape[1] == cfg.n_codebooks + 1


def test_oracle_decode_matches_reference_waveform(case):
    cfg, state, z = case
    orc = D.DacOracle(cfg, state)
    got = orc.from_indices(torch.from_numpy(z["codes"]).clone())
    assert float((got - torch.from_numpy(z["decoded"])).pow(2).mean().sqrt()) <= 1e-6
    rnd = torch.from_numpy(z["rnd_codes"]).clone()
    zq = orc.dequantize(rnd)
    assert np.array_equal(rnd.numpy(), z["rnd_codes_clamped"])  # in-place clamp, rvq.py:354-359
    assert float((zq - torch.from_numpy(z["rnd_z"])).abs().max()) <= 1e-5
    got = orc.from_indices(torch.from_numpy(z["rnd_codes"]).clone())
    assert float((got - torch.from_numpy(z["rnd_decoded"])).pow(2).mean().sqrt()) <= 1e-6
    assert got.shape == (2, 1, 5 * cfg.frame_length) and float(got.abs().max()) < 1.0


@pytest.mark.skipif(not __import__("oracle.refload", fromlist=["x"]).reference_available(),
                    reason="reference checkout not present on this box")
def test_oracle_in_bf16_equals_the_reference_bf16_module():
    """The text2semantic CLI holds the codec as `codec.to(dtype=torch.bfloat16)` (inference.py:416).  The oracle
    built from the bf16-rounded state reproduces the UNMODIFIED reference module in that mode bit for bit -- decode
    and encode -- which is what pins the calibration target of tests/test_dac_gpu.py's bf16-module test."""
    from oracle.gen_golden_dac import build_reference_dac

    cfg = D.small_config()
    state = D.make_synthetic_state(cfg, seed=11)
    ref = build_reference_dac(cfg, state).to(dtype=torch.bfloat16)
    orc = D.DacOracle(cfg, {k: (v.bfloat16() if v.is_floating_point() else v) for k, v in state.items()})
    codes = D.make_codes(cfg, 2, 5, seed=4)
    with torch.no_grad():
        want = ref.from_indices(codes.clone())
    got = orc.from_indices(codes.clone())
    assert want.dtype == got.dtype == torch.bfloat16 and torch.equal(got, want)
    n = 3 * cfg.frame_length - 700
    audio = 0.2 * torch.randn(1, 1, n, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        c_ref, _ = ref.encode(audio.bfloat16(), torch.tensor([n]))
    c_orc, _ = orc.encode(audio.bfloat16(), torch.tensor([n]))
    assert torch.equal(c_orc, c_ref)
    c32, _ = D.DacOracle(cfg, state).encode(audio, torch.tensor([n]))
    print("reference bf16 encode agrees with its own fp32 codes in", float((c_ref == c32).float().mean()), "of the indices")


def test_decoder_is_causal_prefix_consistent(case):
    """All convolutions are causal and attention is windowed-causal: decoding a prefix of the codes
    gives the prefix of the waveform (the property streaming decode relies on)."""
    cfg, state, z = case
    orc = D.DacOracle(cfg, state)
    codes = D.make_codes(cfg, 1, 6, seed=8)
    full = orc.from_indices(codes.clone())
    part = orc.from_indices(codes[:, :, :4].clone())
    assert float((full[..., : 4 * cfg.frame_length] - part).abs().max()) <= 1e-5


def test_state_table_and_weight_norm_fold(case):
    cfg, state, _ = case
    folded = D.fold_weight_norm(state)
    assert not any(k.endswith(("weight_g", "weight_v", "original0", "original1")) for k in folded)
    k = "decoder.model.1.block.1.conv"
    v, g = state[k + ".parametrizations.weight.original1"], state[k + ".parametrizations.weight.original0"]
    want = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    assert torch.allclose(folded[k + ".weight"], want, rtol=1e-6, atol=1e-7)
    full = D.state_shapes(D.DacConfig())
    n_params = sum(int(np.prod(s)) for s in full.values())
    assert 385e6 < n_params < 400e6  # 391.4 M parameters with the yaml's values


def test_product_shim_fold_equals_oracle_fold(case):
    from fish_speech_amd.dac import fold_weight_norm

    cfg, state, _ = case
    a, b = fold_weight_norm(state), D.fold_weight_norm(state)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_decoder_receptive_field_bounds_the_streaming_context():
    """The incremental decode (fmi_dac_decode_tail) runs the decoder conv stack on frames [t0-ctx, T) and
    drops the context's samples.  On the restated reference decoder: with ctx = the receptive field computed
    by the library's formula (19 latent columns -> 5 frames for rates 8,8,4,2; here the small config's) the
    kept samples equal the full decode's, and with one column less of context they do not."""
    cfg = D.small_config()
    state = D.make_synthetic_state(cfg, seed=2)
    orc = D.DacOracle(cfg, state)
    ctx = 6
    for r in reversed(cfg.decoder_rates):
        ctx = -(-(ctx + 6 * (1 + 3 + 9)) // r) + 1
    ctx += 6
    codes = D.make_codes(cfg, 1, 4 + -(-ctx // 4) + 3, seed=4)
    z = orc.dequantize(codes.clone())
    full = orc.decoder(z)
    hop = cfg.hop_length
    col0 = z.shape[-1] - 8                   # keep the audio of the last 8 latent columns
    ok = orc.decoder(z[:, :, col0 - ctx:])[..., ctx * hop:]
    assert torch.allclose(ok, full[..., col0 * hop:], atol=2e-6, rtol=0)
    short = orc.decoder(z[:, :, col0 - ctx + 1:])[..., (ctx - 1) * hop:]
    assert not torch.allclose(short, full[..., col0 * hop:], atol=2e-6, rtol=0)


def test_stream_chunk_schedule():
    from fish_speech_amd.stream import chunk_schedule

    assert chunk_schedule(215, 9, 32) == [9, 41, 73, 105, 137, 169, 201, 215]
    assert chunk_schedule(5, 9, 32) == [5]
    assert chunk_schedule(9, 9, 4) == [9]
    assert chunk_schedule(10, 2, 4) == [2, 6, 10]
    import pytest
    with pytest.raises(ValueError):
        chunk_schedule(10, 0, 4)


def test_extract_vq_sharding_and_cli_surface(tmp_path, monkeypatch):
    """Host logic of the batch encode tool (tools/vqgan/extract_vq.py:143-207): rank slicing r::R over the files
    that have no .npy yet, rank/world discovery, the reference's options."""
    from click.testing import CliRunner

    from fish_speech_amd import extract_vq as X

    for i in range(7):
        (tmp_path / f"f{i}.wav").write_bytes(b"x")
    (tmp_path / "notes.txt").write_text("not audio")
    (tmp_path / "f3.npy").write_bytes(b"x")
    files = X.list_audio_files(str(tmp_path))
    assert [f.name for f in files] == [f"f{i}.wav" for i in range(7)]
    parts = [X.pending_for_rank(files, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == sorted(f for f in files if f.name != "f3.wav")
    assert [f.name for f in parts[0]] == ["f0.wav", "f4.wav"]           # files[rank::world] after the skip
    fl = tmp_path / "list.txt"
    fl.write_text(f"{tmp_path / 'f1.wav'}|spk|en|hello\n\n{tmp_path / 'f2.wav'}|spk|en|x\n")
    assert [p.name for p in X.load_filelist(fl)] == ["f1.wav", "f2.wav"]
    for k in ("SLURM_PROCID", "SLURM_NTASKS", "RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert X._rank_world() == (0, 1)
    monkeypatch.setenv("RANK", "2"), monkeypatch.setenv("WORLD_SIZE", "4")
    assert X._rank_world() == (2, 4)
    monkeypatch.setenv("SLURM_PROCID", "1"), monkeypatch.setenv("SLURM_NTASKS", "8")
    assert X._rank_world() == (1, 8)
    res = CliRunner().invoke(X.main, ["--help"])
    assert res.exit_code == 0
    for flag in ("--num-workers", "--config-name", "--checkpoint-path", "--batch-size", "--filelist", "FOLDER"):
        assert flag in res.output


def test_third_party_dac_restatement_agrees_with_the_transformers_port():
    """The codec imports `dac.nn.quantize` / `dac.nn.layers` from descript-audio-codec, a wheel that
    is not in this image; oracle/stubs/dac restates them.  HF transformers ships an independent port of the same
    package (`transformers.models.dac.modeling_dac`), so the restatement is checked against it with shared weights:
    Snake1d bit for bit, `ResidualVectorQuantize.from_codes` bit for bit, the whole residual quantisation loop
    (in_proj -> nearest normalised code -> out_proj -> subtract) with equal codes and bit-equal sums.  The port writes
    the distance as -(|e|^2 - 2 e.c) + |c|^2 where the package has -(|e|^2 - 2 e.c + |c|^2): |c|^2 is 1 up to rounding
    for the l2-normalised codebook, so the argmax can only differ on near-ties -- none on this input."""
    import sys

    import pytest as _pytest

    modeling = _pytest.importorskip("transformers.models.dac.modeling_dac")
    from transformers.models.dac.configuration_dac import DacConfig

    from oracle.refload import STUBS

    if STUBS not in sys.path:
        sys.path.insert(0, STUBS)
    from dac.nn.layers import Snake1d
    from dac.nn.quantize import ResidualVectorQuantize

    torch.manual_seed(0)
    # Snake1d (dac/nn/layers.py): x + sin^2(alpha x) / (alpha + 1e-9)
    mine, theirs = Snake1d(24), modeling.Snake1d(24)
    with torch.no_grad():
        mine.alpha.copy_(torch.rand(1, 24, 1) * 3 + 0.05)
        theirs.alpha.copy_(mine.alpha)
    x = torch.randn(2, 24, 77) * 2
    assert torch.equal(mine(x), theirs(x))

    D, NCB, CBS, CD = 48, 4, 256, 8
    rvq = ResidualVectorQuantize(input_dim=D, n_codebooks=NCB, codebook_size=CBS, codebook_dim=CD).eval()
    cfg = DacConfig(hidden_size=D, n_codebooks=NCB, codebook_size=CBS, codebook_dim=CD, quantizer_dropout=0.0)
    port = modeling.DacResidualVectorQuantizer(cfg).eval()
    z = torch.randn(2, D, 61)
    with torch.no_grad():
        rvq(z)                                   # materialises the weight-normed convs' effective weights
        for q, p in zip(rvq.quantizers, port.quantizers):
            p.in_proj.weight.copy_(q.in_proj.weight); p.in_proj.bias.copy_(q.in_proj.bias)
            p.out_proj.weight.copy_(q.out_proj.weight); p.out_proj.bias.copy_(q.out_proj.bias)
            p.codebook.weight.copy_(q.codebook.weight)
        zq, codes, latents, _, _ = rvq(z)
        pq, pcodes, platents, _, _ = port(z)
        assert torch.equal(codes, pcodes) and codes.shape == (2, NCB, 61)
        assert torch.equal(latents, platents) and torch.equal(zq, pq)
        a, b, c = rvq.from_codes(codes)
        pa, pb, pc = port.from_codes(codes)
        assert torch.equal(a, pa) and torch.equal(b, pb) and torch.equal(c, pc)

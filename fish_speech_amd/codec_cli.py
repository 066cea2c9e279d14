"""Codec CLI with the flags of the reference's `fish_speech/models/dac/inference.py:50-122`
(-i / -o / --checkpoint-path / -d; --config-name accepted and ignored: the yaml's values are the
defaults of DacConfig).  wav in -> codes .npy + reconstructed wav; .npy in -> wav.

torchaudio / soundfile are not in this image: audio I/O uses scipy.io.wavfile (16-bit or float wav)
and a polyphase resampler, which is host plumbing outside the hot path."""
from __future__ import annotations

from pathlib import Path

import click
import numpy as np
import torch


def _load_wav(path: Path, sample_rate: int) -> torch.Tensor:
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    sr, data = wavfile.read(str(path))
    x = data.astype(np.float32)
    if np.issubdtype(data.dtype, np.integer):
        x /= float(np.iinfo(data.dtype).max)
    if x.ndim == 2:
        x = x.mean(axis=1)  # mono mean, dac/inference.py:79-80
    if sr != sample_rate:
        g = np.gcd(sr, sample_rate)
        x = resample_poly(x, sample_rate // g, sr // g).astype(np.float32)
    return torch.from_numpy(x)[None, None]


@click.command()
@click.option("--input-path", "-i", default="test.wav", type=click.Path(exists=True, path_type=Path))
@click.option("--output-path", "-o", default="fake.wav", type=click.Path(path_type=Path))
@click.option("--config-name", default="modded_dac_vq")
@click.option("--checkpoint-path", default="checkpoints/openaudio-s1-mini/codec.pth")
@click.option("--device", "-d", default="cuda")
def main(input_path, output_path, config_name, checkpoint_path, device):
    from scipy.io import wavfile

    from .dac import MiDAC

    dev = "cuda:0" if device == "cuda" else device
    model = MiDAC.from_checkpoint(checkpoint_path, device=dev)
    if input_path.suffix == ".npy":
        indices = torch.from_numpy(np.load(input_path)).to(dev).long()
        assert indices.ndim == 2, f"Expected 2D indices, got {indices.ndim}"
    else:
        audio = _load_wav(input_path, model.sample_rate).to(dev)
        lengths = torch.tensor([audio.shape[2]], device=dev, dtype=torch.long)
        indices, _ = model.encode(audio, lengths)
        indices = indices[0]
        np.save(output_path.with_suffix(".npy"), indices.cpu().numpy())
    fake = model.from_indices(indices[None].clone())
    wavfile.write(str(output_path), model.sample_rate, fake[0, 0].float().cpu().numpy())
    print(f"{indices.shape[1]} frames -> {fake.shape[-1] / model.sample_rate:.2f} s written to {output_path}")


if __name__ == "__main__":
    main()

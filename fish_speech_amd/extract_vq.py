"""Batch codec encode of a folder of audio files -> one `<file>.npy` of codes (1+n_codebooks, T) int64 per file:
the tool of tools/vqgan/extract_vq.py, with its argument and options.

Sharding is the reference's (`files[RANK::WORLD_SIZE]`, extract_vq.py:207): independent files, one process per
GPU, no collective.  RANK / WORLD_SIZE come from SLURM (as upstream), torchrun's RANK / WORLD_SIZE, or
`--num-workers N`, which spawns N copies of this command, one per visible GPU (extract_vq.py:162-195).
Files whose .npy already exists are skipped; unreadable files are reported and skipped (extract_vq.py:104-111,203).
Audio I/O: wav via scipy (torchaudio is not in this image); mono mean + resample to the codec rate on the host."""
from __future__ import annotations

import os
import subprocess as sp
import sys
import time
from datetime import timedelta
from pathlib import Path
from typing import List, Sequence, Tuple

import click
import numpy as np
import torch

AUDIO_EXTENSIONS = {".mp3", ".wav", ".flac", ".ogg", ".m4a", ".wma", ".aac", ".aiff", ".aif", ".aifc"}  # utils/file.py


def _rank_world() -> Tuple[int, int]:
    for r, w in (("SLURM_PROCID", "SLURM_NTASKS"), ("RANK", "WORLD_SIZE")):
        if r in os.environ and w in os.environ:
            return int(os.environ[r]), int(os.environ[w])
    return 0, 1


def list_audio_files(folder: str) -> List[Path]:
    return sorted(p for p in Path(folder).rglob("*") if p.is_file() and p.suffix.lower() in AUDIO_EXTENSIONS)


def load_filelist(path: Path) -> List[Path]:
    """`path|speaker|lang|text` lines (fish_speech/utils/file.py:load_filelist): only the path column is used."""
    out = []
    for line in Path(path).read_text(encoding="utf-8").splitlines():
        if line.strip():
            out.append(Path(line.split("|")[0]))
    return out


def pending_for_rank(files: Sequence[Path], rank: int, world: int) -> List[Path]:
    todo = [Path(f) for f in files if not Path(f).with_suffix(".npy").exists()]
    return todo[rank::world]


@torch.inference_mode()
def process_batch(files: Sequence[Path], model) -> float:
    """Encode one batch of files and write their .npy; returns the seconds of audio processed
    (extract_vq.py:93-140)."""
    from .codec_cli import _load_wav

    wavs, kept = [], []
    for f in files:
        try:
            wavs.append(_load_wav(f, model.sample_rate)[0, 0])
            kept.append(f)
        except Exception as e:
            print(f"Error reading {f}: {e}", file=sys.stderr)
    if not kept:
        return 0.0
    lengths = [int(w.numel()) for w in wavs]
    longest = max(lengths)
    audios = torch.stack([torch.nn.functional.pad(w, (0, longest - w.numel())) for w in wavs])[:, None].to(model.device)
    indices, feature_lengths = model.encode(audios, torch.tensor(lengths, device=model.device, dtype=torch.long))
    out = indices.cpu().numpy()
    for f, n, feat in zip(kept, feature_lengths.cpu().tolist(), out):
        with open(f.with_suffix(".npy"), "wb") as fh:
            np.save(fh, feat[:, :n])
    return sum(lengths) / model.sample_rate


def get_model(checkpoint_path: str, device: str = "cuda:0"):
    from .dac import MiDAC

    state = torch.load(checkpoint_path, map_location="cpu", mmap=True, weights_only=True)
    return MiDAC(device=device).load_state_dict(state)


@click.command()
@click.argument("folder")
@click.option("--num-workers", default=1)
@click.option("--config-name", default="modded_dac_vq")
@click.option("--checkpoint-path", default="checkpoints/s2-pro/codec.pth")
@click.option("--batch-size", default=64)
@click.option("--filelist", default=None, type=Path)
def main(folder, num_workers, config_name, checkpoint_path, batch_size, filelist):
    rank, world = _rank_world()
    if num_workers > 1 and world != num_workers:
        assert world == 1, "You should either use SLURM or this launcher, not both"
        visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
        devices = visible.split(",") if visible else [str(i) for i in range(max(torch.cuda.device_count(), 1))]
        procs = []
        for i in range(num_workers):
            env = os.environ.copy()
            env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = devices[i % len(devices)]
            env["SLURM_PROCID"], env["SLURM_NTASKS"] = str(i), str(num_workers)
            procs.append(sp.Popen([sys.executable, "-m", "fish_speech_amd.extract_vq"] + sys.argv[1:], env=env))
        for p in procs:
            p.wait()
        print("All workers finished")
        return
    files = load_filelist(filelist) if filelist else list_audio_files(folder)
    print(f"Found {len(files)} files")
    mine = pending_for_rank(files, rank, world)
    print(f"[rank {rank}/{world}] processing {len(mine)} files")
    model = get_model(checkpoint_path)
    t0, audio_s, done = time.time(), 0.0, 0
    for n_batch, i in enumerate(range(0, len(mine), batch_size)):
        batch = mine[i:i + batch_size]
        audio_s += process_batch(batch, model)
        done += len(batch)
        if (n_batch + 1) % 10 == 0:
            eta = (time.time() - t0) / done * (len(mine) - done)
            print(f"Processed {done} files, {audio_s / 3600:.2f} hours of audio, ETA: {timedelta(seconds=round(eta))}")
    print(f"Finished processing {len(mine)} files, {audio_s / 3600:.2f} hours of audio")


if __name__ == "__main__":
    main()

"""Batch codec helpers of the API server (reference: tools/server/model_utils.py:15-86) over MiDAC.

* ``batch_encode``      -- same contract: list of audio byte strings (wav) or (1, n) tensors -> list of per-item code
                           tensors (1 + n_codebooks, T_i), trimmed to each item's frame count.
* ``batch_vqgan_decode`` -- the reference's version calls ``model.decode(padded, feature_lengths=...)``, a signature the
                           current DAC no longer has (stale); the behaviour it was written for is
                           restated against the codec's real entry point: right-pad the code matrices to the longest,
                           decode in micro-batches of MICRO_BATCH_SIZE through ``from_indices``, trim every waveform
                           to ``T_i * frame_length`` samples.  All layers are causal, so right padding never changes an
                           item's own samples.
* ``cached_vqgan_batch_encode`` -- LRU over (device, audio bytes), like the reference's cachetools wrapper.

The reference wraps both in ``torch.autocast(dtype=torch.half)``; MiDAC encodes in fp32 (codes stay bit-comparable) and
decodes in its configured precision -- fp16 is not implemented."""
from __future__ import annotations

import collections
import io
from typing import List, Sequence, Union

import numpy as np
import torch

CACHE_MAXSIZE = 10000
MICRO_BATCH_SIZE = 8


def _decode_wav_bytes(data: bytes, sample_rate: int) -> torch.Tensor:
    """wav bytes -> (1, n) float32 mono at `sample_rate` (the reference uses librosa.load(sr=...), absent here)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    sr, x = wavfile.read(io.BytesIO(data))
    y = x.astype(np.float32)
    if np.issubdtype(x.dtype, np.integer):
        y /= float(np.iinfo(x.dtype).max)
    if y.ndim == 2:
        y = y.mean(axis=1)
    if sr != sample_rate:
        g = np.gcd(sr, sample_rate)
        y = resample_poly(y, sample_rate // g, sr // g).astype(np.float32)
    return torch.from_numpy(y)[None]


@torch.no_grad()
def batch_encode(model, audios_list: Sequence[Union[bytes, torch.Tensor]]) -> List[torch.Tensor]:
    """model_utils.py:15-49."""
    sample_rate = model.spec_transform.sample_rate if hasattr(model, "spec_transform") else model.sample_rate
    audios = [_decode_wav_bytes(a, sample_rate) if isinstance(a, (bytes, bytearray)) else a for a in audios_list]
    lengths = torch.tensor([a.shape[-1] for a in audios], device=model.device)
    max_length = int(lengths.max().item())
    padded = torch.stack([torch.nn.functional.pad(a.float(), (0, max_length - a.shape[-1])) for a in audios]).to(model.device)
    features, feature_lengths = model.encode(padded, audio_lengths=lengths)
    features, feature_lengths = features.cpu(), feature_lengths.cpu()
    return [f[..., : int(n)] for f, n in zip(features, feature_lengths)]


_cache: "collections.OrderedDict" = collections.OrderedDict()


def cached_vqgan_batch_encode(model, audios: Sequence[bytes]) -> List[torch.Tensor]:
    """model_utils.py:52-57: LRU keyed by (device, tuple of the audio byte strings)."""
    key = (str(model.device), tuple(audios))
    if key in _cache:
        _cache.move_to_end(key)
        return _cache[key]
    out = batch_encode(model, audios)
    _cache[key] = out
    while len(_cache) > CACHE_MAXSIZE:
        _cache.popitem(last=False)
    return out


@torch.no_grad()
def batch_vqgan_decode(model, features: Sequence[torch.Tensor]) -> List[np.ndarray]:
    """model_utils.py:60-86 (see the module docstring for the stale call it replaces).  features[i]: integer
    (1 + n_codebooks, T_i).  Returns float32 arrays (1, T_i * frame_length)."""
    lengths = [int(f.shape[-1]) for f in features]
    max_length = max(lengths)
    padded = torch.stack([torch.nn.functional.pad(f.long(), (0, max_length - f.shape[-1])) for f in features]).to(model.device)
    audios = []
    for i in range(0, padded.shape[0], MICRO_BATCH_SIZE):
        audios.append(model.from_indices(padded[i: i + MICRO_BATCH_SIZE].contiguous()).float().cpu())
    audios = torch.cat(audios, dim=0)
    fl = model.frame_length
    return [a[..., : n * fl].numpy() for a, n in zip(audios, lengths)]

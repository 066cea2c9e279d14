"""fish_speech_amd -- MI355X (gfx950) native hot path of fish-speech S2 inference.

Host-side mirrors of the reference's duck-typed seams:
  * ``MiDualAR`` / ``decode_one_token`` / ``generate``  <->  DualARTransformer, decode_one_token_ar,
    generate (fish_speech/models/text2semantic/{llama,inference}.py)
  * ``MiDAC``  <->  DAC.encode / DAC.from_indices (fish_speech/models/dac/modded_dac.py)
All compute runs in hand-written HIP kernels behind the C ABI of ``include/fishmi.h``;
PyTorch is used for device memory, streams and torch.distributed only.
"""
from ._lib import FishmiError, load as load_library  # noqa: F401

__all__ = ["FishmiError", "load_library"]

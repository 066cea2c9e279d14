"""Test-only stand-in for `descript-audiotools` 0.7.2 (reference uv.lock:893-894): modded_dac.py
imports AudioSignal (unused at inference) and ml.BaseModel (an nn.Module with a .device)."""


class AudioSignal:  # never constructed on the inference path
    pass

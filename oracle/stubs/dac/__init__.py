"""Test-only restatement of the parts of `descript-audio-codec` 1.0.0 (pinned by the reference's
uv.lock:864-865) that fish_speech/models/dac/{modded_dac,rvq}.py import.  The real package is not in
this image and cannot be fetched, so these few functions are restated from the published source
and could NOT be diffed against the wheel: parity of this third-party arithmetic is UNPINNED against the package
itself (see oracle/README.md); it IS cross-checked, bit for bit, against HF transformers' independent port of the
package (tests/test_dac_cpu.py).  Used only to import the unmodified reference for golden generation."""

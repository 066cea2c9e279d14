"""Test-only restatement of the parts of `descript-audio-codec` 1.0.0 (pinned by the reference's
uv.lock:864-865) that fish_speech/models/dac/{modded_dac,rvq}.py import.  The real package is not in
this image and cannot be fetched, so these few functions are restated from the published source
and could NOT be diffed against the wheel: parity of this third-party arithmetic is UNPINNED
(see oracle/README.md).  Used only to import the unmodified reference for golden generation."""

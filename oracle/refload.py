"""Helpers to import the UNMODIFIED reference modules (authoring container only).

Test infrastructure: used by oracle/gen_golden*.py and tests/test_oracle_cpu.py / tests/test_dac_cpu.py.  Nothing
here is reachable from the GPU-side tests, smoke() or bench.py (/root/reference does not exist on
the GPU box)."""
from __future__ import annotations

import os
import sys

REFERENCE_ROOT = os.environ.get("FISH_REFERENCE_ROOT", "/root/reference")
STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "fish_speech"))


def add_reference_to_path():
    if not reference_available():
        raise RuntimeError("reference checkout not present")
    for p in (REFERENCE_ROOT, STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)


class FakeTokenizer:
    """generate() only needs tokenizer.get_token_id('<|im_end|>') (inference.py:205,320)."""

    def __init__(self, im_end_id: int):
        self.im_end_id = im_end_id

    def get_token_id(self, token: str) -> int:
        assert token == "<|im_end|>", token
        return self.im_end_id


def build_reference_dual_ar(cfg, state):
    """Instantiate the reference DualARTransformer from an oracle config + state dict."""
    add_reference_to_path()
    import torch
    from fish_speech.models.text2semantic.llama import DualARModelArgs, DualARTransformer

    args = DualARModelArgs(**cfg.reference_kwargs())
    model = DualARTransformer(args)
    dtype = state["embeddings.weight"].dtype
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(m in ("freqs_cis", "causal_mask", "fast_freqs_cis") for m in missing), missing
    model = model.to(dtype=dtype).eval()
    model.tokenizer = FakeTokenizer(cfg.im_end_id)
    model._cache_setup_done = False
    return model

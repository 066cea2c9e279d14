"""Test-only stand-in for `loralib` (absent from this image). The inference path never builds LoRA layers."""


class _Missing:
    def __init__(self, *a, **k):
        raise RuntimeError("loralib stub: LoRA is not part of the inference hot path")


Linear = Embedding = MergedLinear = _Missing


def mark_only_lora_as_trainable(*a, **k):
    raise RuntimeError("loralib stub")

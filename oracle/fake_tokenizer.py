"""A deterministic stand-in for the HF tokenizer behind fish_speech.tokenizer.FishTokenizer (no tokenizer files
exist in the authoring container).  TEST INFRASTRUCTURE: used by oracle/gen_golden_prompt.py to drive the
UNMODIFIED reference prompt classes and by tests/ to drive ours with the very same token ids.

Vocabulary: ids 0..255 = UTF-8 bytes; specials follow in the order of fish_speech/tokenizer.py:37-50
(<|endoftext|>, <|pad|>, <|im_start|>, <|im_end|>, ..., <|audio_pad|>), then <|speaker:0..15|>, then the 4096
contiguous <|semantic:i|> ids.  `encode` splits on the `<|...|>` specials like a tokenizer with
`allowed_special="all"` (tokenizer.py:104-113) and byte-encodes everything else."""
from __future__ import annotations

import re
from typing import List

SPECIALS = ["<|endoftext|>", "<|pad|>", "<|im_start|>", "<|im_end|>", "<|phoneme_start|>", "<|phoneme_end|>",
            "<|text|>", "<|voice|>", "<|interleave|>", "<|audio_start|>", "<|audio_end|>", "<|audio_pad|>"]
SPECIALS += [f"<|speaker:{i}|>" for i in range(16)]
N_SEMANTIC = 4096


class ByteTokenizer:
    def __init__(self):
        self.vocab = {tok: 256 + i for i, tok in enumerate(SPECIALS)}
        self.semantic_begin_id = 256 + len(SPECIALS)
        self.semantic_end_id = self.semantic_begin_id + N_SEMANTIC - 1
        for i in range(N_SEMANTIC):
            self.vocab[f"<|semantic:{i}|>"] = self.semantic_begin_id + i
        self.inv = {v: k for k, v in self.vocab.items()}
        self._pat = re.compile(r"(<\|[a-z_]+(?::\d+)?\|>)")

    @property
    def vocab_size(self) -> int:
        return self.semantic_end_id + 1

    def get_token_id(self, token: str) -> int:
        return self.vocab[token]

    def encode(self, text: str, add_special_tokens: bool = False, **kw) -> List[int]:
        out: List[int] = []
        for piece in self._pat.split(text):
            if not piece:
                continue
            if piece in self.vocab:
                out.append(self.vocab[piece])
            else:
                out.extend(piece.encode("utf-8"))
        return out

    def decode(self, tokens, **kw) -> str:
        if isinstance(tokens, int):
            tokens = [tokens]
        buf, out = bytearray(), []
        for t in tokens:
            if t < 256:
                buf.append(t)
            else:
                out.append(buf.decode("utf-8", "replace"))
                buf = bytearray()
                out.append(self.inv.get(int(t), f"<{int(t)}>"))
        out.append(buf.decode("utf-8", "replace"))
        return "".join(out)

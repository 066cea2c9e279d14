"""Prompt builder: conversation -> the (1 + num_codebooks, T) integer prompt the Dual-AR model consumes.

Mirrors what the hot path is fed by the reference (fish_speech/conversation.py:33-103 `Conversation`,
fish_speech/content_sequence.py:154-324 `ContentSequence.encode / encode_for_inference`), as one flat pass:
every message is lowered to a list of segments -- literal token ids or VQ code blocks -- and the segments are
laid into the prompt matrix directly (row 0: token ids, rows 1..: codebook indices under VQ columns, zero
elsewhere).  Loss masks / label shifting (training-only fields of the reference's EncodedMessage) are not built.

The tokenizer is duck-typed like the reference uses it (fish_speech/tokenizer.py:55-129): `encode(text,
add_special_tokens=False) -> list[int]` and `semantic_begin_id` (VQ code c of codebook 0 is token
`semantic_begin_id + c`, content_sequence.py:214-223: semantic ids are contiguous).
Parity: tests/test_prompt_cpu.py compares with prompts produced by the unmodified reference classes
(tests/golden/prompt_cases.json, written by oracle/gen_golden_prompt.py)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

IM_START_TOKEN = "<|im_start|>"            # fish_speech/tokenizer.py:15-16
IM_END_TOKEN = "<|im_end|>"
MODALITY_TOKENS = {"text": "<|text|>", "voice": "<|voice|>", "interleave": "<|interleave|>"}  # tokenizer.py:27-31


@dataclass
class TextPart:
    """Literal text (tokenised with special tokens recognised inline) or ready token ids."""
    text: Optional[str] = None
    tokens: Optional[Sequence[int]] = None
    cal_loss: bool = False

    def __post_init__(self):
        if self.text is None and self.tokens is None:  # content_sequence.py:47-49
            raise ValueError("Either text or tokens must be provided")


@dataclass
class VQPart:
    """Codec frames: codes (num_codebooks, n) integer tensor / array."""
    codes: Union[torch.Tensor, np.ndarray]
    cal_loss: bool = False


@dataclass
class Message:
    role: str                                   # "system" | "user" | "assistant"
    parts: List[Union[TextPart, VQPart]] = field(default_factory=list)
    add_im_start: bool = True
    add_im_end: bool = True
    cal_loss: bool = False
    modality: Optional[str] = None              # "text" | "voice" | "interleave"
    ignore_im_start_loss: bool = True


class Conversation:
    def __init__(self, messages: Optional[List[Message]] = None):
        self.messages: List[Message] = list(messages) if messages else []

    def append(self, message: Message):
        self.messages.append(message)

    def copy(self) -> "Conversation":
        """Messages are treated as immutable once appended, so a shallow copy of the list is a snapshot."""
        return Conversation(self.messages)

    # ---- lowering
    def _segments(self):
        """-> list of ("text", str) | ("ids", sequence) | ("vq", int64 array (ncb, n)) in prompt order."""
        seg = []
        for m in self.messages:
            if m.add_im_start:  # conversation.py:50-60: "<|im_start|>{role}\n{modality token}"
                seg.append(("text", f"{IM_START_TOKEN}{m.role}\n{MODALITY_TOKENS[m.modality] if m.modality else ''}"))
            for p in m.parts:
                if isinstance(p, TextPart):
                    seg.append(("ids", list(p.tokens)) if p.tokens is not None else ("text", p.text))
                elif isinstance(p, VQPart):
                    c = p.codes.detach().cpu().numpy() if isinstance(p.codes, torch.Tensor) else np.asarray(p.codes)
                    seg.append(("vq", c.astype(np.int32).astype(np.int64)))  # `.to(torch.int)`, content_sequence.py:219
                else:  # the reference's encode rejects anything else too (content_sequence.py:227-228)
                    raise ValueError(f"Unsupported part type: {type(p)}")
            if m.add_im_end:    # conversation.py:72-76
                seg.append(("text", IM_END_TOKEN + "\n"))
        return seg

    def encode_for_inference(self, tokenizer, num_codebooks: int):
        """-> (values int64 (1+num_codebooks, T), None, None), like conversation.py:94-101.
        (The reference's audio_masks / audio_parts are only ever non-None for AudioPart inputs, which its own
        encode() rejects; the Dual-AR forward ignores them, llama.py:423-433.)"""
        cols_tok: List[np.ndarray] = []
        cols_vq: List[Optional[np.ndarray]] = []
        for kind, payload in self._segments():
            if kind == "text":
                ids = np.asarray(tokenizer.encode(payload, add_special_tokens=False), dtype=np.int64)
                cols_tok.append(ids)
                cols_vq.append(None)
            elif kind == "ids":
                cols_tok.append(np.asarray(payload, dtype=np.int64))
                cols_vq.append(None)
            else:
                if payload.shape[0] != num_codebooks:
                    raise ValueError(f"VQ part has {payload.shape[0]} codebooks, expected {num_codebooks}")
                cols_tok.append(payload[0] + int(tokenizer.semantic_begin_id))
                cols_vq.append(payload)
        T = int(sum(len(c) for c in cols_tok))
        values = np.zeros((num_codebooks + 1, T), dtype=np.int64)
        at = 0
        for ids, vq in zip(cols_tok, cols_vq):
            n = len(ids)
            values[0, at:at + n] = ids
            if vq is not None:
                values[1:, at:at + n] = vq
            at += n
        return torch.from_numpy(values), None, None

    def visualize(self, tokenizer, **_):
        """Plain-text rendering of the prompt (the reference colour-codes loss tokens; there is no loss here)."""
        ncb = max([np.asarray(p.codes).shape[0] for m in self.messages for p in m.parts if isinstance(p, VQPart)] or [1])
        values, _, _ = self.encode_for_inference(tokenizer, num_codebooks=ncb)
        print(tokenizer.decode(values[0].tolist()))

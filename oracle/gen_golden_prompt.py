"""Writes tests/golden/prompt_cases.json: prompts built by the UNMODIFIED reference (fish_speech.conversation,
fish_speech.content_sequence) and the call trace of its generate_long (text2semantic/inference.py:523-733) --
which prompts it hands to `generate`, which codes it yields -- with `generate` stubbed by a deterministic function
and the tokenizer replaced by oracle/fake_tokenizer.ByteTokenizer (no tokenizer files exist here).

Run in the authoring container only:  python -m oracle.gen_golden_prompt"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

from oracle.fake_tokenizer import ByteTokenizer
from oracle.refload import add_reference_to_path

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "prompt_cases.json")
NCB = 10


def codes_for(seed: int, n: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(0, 1024, (NCB, n), generator=g)
    c[0] = torch.randint(0, 4096, (n,), generator=g)
    return c


def stub_generate_factory(tok, log):
    """Deterministic replacement of inference.generate: the number of new frames and the codes depend only on
    the prompt length, so the reference's and our generate_long can be driven identically."""

    def generate(*, model, prompt, max_new_tokens, **kw):
        T = int(prompt.shape[1])
        n = 3 + T % 5
        new = torch.zeros(NCB + 1, n + 1, dtype=prompt.dtype)
        c = codes_for(T, n)
        new[1:, :n] = c
        new[0, :n] = c[0] + tok.semantic_begin_id
        new[0, n] = tok.get_token_id("<|im_end|>")          # last column: the <|im_end|> frame (dropped by [:-1])
        log.append({"prompt": prompt.cpu().numpy().tolist(), "max_new_tokens": int(max_new_tokens),
                    "temperature": float(kw.get("temperature", -1)), "top_p": float(kw.get("top_p", -1)),
                    "top_k": int(kw.get("top_k", -1))})
        return torch.cat([prompt.cpu(), new], dim=1)

    return generate


CONVERSATIONS = [
    # (name, list of messages); a part is ["text", str] | ["tokens", [ids]] | ["vq", seed, n_frames]
    ("plain_tts", [
        {"role": "system", "parts": [["text", "convert the provided text to speech"]]},
        {"role": "user", "parts": [["text", "<|speaker:0|>Hello there, General Kenobi."]]},
        {"role": "assistant", "parts": [], "modality": "voice", "add_im_end": False},
    ]),
    ("voice_clone", [
        {"role": "system", "parts": [["text", "convert the provided text to speech reference to the following:\n\nText:\n"],
                                     ["text", "<|speaker:0|>A reference line.\n<|speaker:1|>Zweite Zeile: äöü ✓"],
                                     ["text", "\n\nSpeech:\n"], ["vq", 7, 23]]},
        {"role": "user", "parts": [["text", "<|speaker:1|>Now say this."]]},
        {"role": "assistant", "parts": [], "modality": "voice", "add_im_end": False},
    ]),
    ("multi_turn", [
        {"role": "system", "parts": [["text", "convert the provided text to speech"]]},
        {"role": "user", "parts": [["text", "<|speaker:0|>First chunk."]]},
        {"role": "assistant", "parts": [["vq", 11, 9]], "modality": "voice"},
        {"role": "user", "parts": [["text", "<|speaker:0|>Second chunk, 日本語も."], ["tokens", [65, 66, 67]]]},
        {"role": "assistant", "parts": [["vq", 12, 1], ["vq", 13, 4]], "modality": "interleave"},
        {"role": "user", "parts": [["text", ""]], "add_im_start": False},
        {"role": "assistant", "parts": [], "modality": "text", "add_im_end": False},
    ]),
    ("empty", []),
]

LONG_CASES = [
    dict(name="no_prompt_two_chunks", text="<|speaker:0|>One short turn.<|speaker:1|>Another speaker answers with a "
         "longer sentence that pushes the byte budget.<|speaker:0|>And a third turn.", chunk_length=60,
         max_new_tokens=64, temperature=0.7, top_p=0.8, top_k=20, num_samples=1),
    dict(name="untagged_text", text="no speaker tags at all, just text", chunk_length=300, max_new_tokens=0,
         temperature=1.0, top_p=0.9, top_k=30, num_samples=2),
    dict(name="voice_clone_prompt", text="<|speaker:0|>Cloned voice line one.<|speaker:0|>Line two.<|speaker:0|>Three."
         "<|speaker:0|>Four.<|speaker:0|>Five.<|speaker:0|>Six turns force a speaker-count split.", chunk_length=512,
         max_new_tokens=32, temperature=0.9, top_p=0.7, top_k=30, num_samples=1,
         prompt_text=["reference without a tag", "<|speaker:3|>tagged reference"], prompt_vq=[[21, 6], [22, 5]]),
]


def main():
    add_reference_to_path()
    import fish_speech.models.text2semantic.inference as RI
    from fish_speech.content_sequence import TextPart, VQPart
    from fish_speech.conversation import Conversation, Message

    tok = ByteTokenizer()
    out = {"num_codebooks": NCB, "conversations": [], "generate_long": [], "split": []}

    def build(parts):
        res = []
        for p in parts:
            if p[0] == "text":
                res.append(TextPart(text=p[1]))
            elif p[0] == "tokens":
                res.append(TextPart(tokens=list(p[1])))
            else:
                res.append(VQPart(codes=codes_for(p[1], p[2])))
        return res

    for name, msgs in CONVERSATIONS:
        conv = Conversation()
        for m in msgs:
            kw = {k: v for k, v in m.items() if k not in ("role", "parts")}
            conv.append(Message(role=m["role"], parts=build(m["parts"]), **kw))
        values, masks, parts = conv.encode_for_inference(tok, num_codebooks=NCB)
        assert masks is None and parts is None
        out["conversations"].append({"name": name, "messages": msgs, "values": values.numpy().tolist()})

    for text, ms, mb in [("<|speaker:0|>a<|speaker:1|> b c <|speaker:12|>", 3, 300), ("plain", 3, 300),
                         ("<|speaker:0|>" + "x" * 50 + "<|speaker:1|>" + "y" * 50 + "<|speaker:0|>z", 5, 80),
                         ("lead in <|speaker:2|>t1<|speaker:2|>t2<|speaker:2|>t3<|speaker:2|>t4", 2, 1000)]:
        turns = RI.split_text_by_speaker(text)
        out["split"].append({"text": text, "max_speakers": ms, "max_bytes": mb, "turns": turns,
                             "batches": RI.group_turns_into_batches(turns, max_speakers=ms, max_bytes=mb)})

    class Cfg:
        num_codebooks = NCB
        max_seq_len = 4096

    class FakeModel:
        config = Cfg()
        tokenizer = tok

        def parameters(self):
            return iter([torch.nn.Parameter(torch.zeros(1))])

    for case in LONG_CASES:
        log = []
        RI.generate = stub_generate_factory(tok, log)
        kw = {k: v for k, v in case.items() if k not in ("name", "prompt_vq")}
        if "prompt_vq" in case:
            kw["prompt_tokens"] = [codes_for(s, n) for s, n in case["prompt_vq"]]
        responses = []
        for r in RI.generate_long(model=FakeModel(), device="cpu", decode_one_token=None, **kw):
            responses.append({"action": r.action, "text": r.text,
                              "codes": None if r.codes is None else r.codes.cpu().numpy().tolist()})
        out["generate_long"].append({"case": case, "calls": log, "responses": responses})

    # the "prompt too long" error of generate_long (inference.py:658-661)
    class SmallCfg(Cfg):
        max_seq_len = 2048 + 40
    FakeModel.config = SmallCfg()
    RI.generate = stub_generate_factory(tok, [])
    try:
        list(RI.generate_long(model=FakeModel(), device="cpu", decode_one_token=None, text="<|speaker:0|>" + "w" * 64))
        err = None
    except ValueError as e:
        err = str(e)
    out["too_long_error"] = err

    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out["conversations"]), "conversations,",
          sum(len(g["calls"]) for g in out["generate_long"]), "generate calls; error:", err)


if __name__ == "__main__":
    sys.exit(main())

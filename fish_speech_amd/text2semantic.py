"""Host orchestration of text -> semantic codes above the Dual-AR seam, mirroring the reference's
fish_speech/models/text2semantic/inference.py for SURVEY.md rows a15-a16:

* `split_text_by_speaker`, `group_turns_into_batches`          (inference.py:454-520)
* `generate_long`  -- prompt assembly, text chunking, multi-turn context carry, `codes = y[1:, T:-1]`,
                      `GenerateResponse("sample" | "next")` stream                    (inference.py:523-733)
* `init_model`, `launch_thread_safe_queue`, `GenerateRequest`, `WrappedGenerateResponse` (inference.py:362-392, 736-799)
* `main` -- the CLI with the reference's flags                                         (inference.py:802-960)

All GPU work goes through `fish_speech_amd.dual_ar` (C ABI -> HIP kernels); nothing here touches the oracle.
The call trace of `generate_long` (which prompts reach `generate`, which codes come back out) is pinned against
the unmodified reference in tests/test_prompt_cpu.py."""
from __future__ import annotations

import os
import queue
import re
import threading
import time
import traceback
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch

from .prompt import Conversation, Message, TextPart, VQPart

_SPEAKER_TAG = re.compile(r"<\|speaker:\d+\|>")


@dataclass
class GenerateResponse:              # inference.py:447-451
    action: str                      # "sample" | "next"
    codes: Optional[torch.Tensor] = None
    text: Optional[str] = None


@dataclass
class WrappedGenerateResponse:       # inference.py:736-739
    status: str                      # "success" | "error"
    response: Optional[Union[GenerateResponse, Exception]] = None


@dataclass
class GenerateRequest:               # inference.py:742-745
    request: dict
    response_queue: queue.Queue


def split_text_by_speaker(text: str) -> List[str]:
    """Turns = each `<|speaker:N|>` tag with the text up to the next tag, stripped; text before the first tag
    is dropped; no tag -> no turns (inference.py:454-484)."""
    turns = []
    tags = list(_SPEAKER_TAG.finditer(text))
    for i, m in enumerate(tags):
        end = tags[i + 1].start() if i + 1 < len(tags) else len(text)
        turns.append((m.group(0) + text[m.end():end]).strip())
    return turns


def group_turns_into_batches(turns: Sequence[str], max_speakers: int = 3, max_bytes: int = 300) -> List[str]:
    """Greedy packing: a batch closes when it already holds `max_speakers` turns or the next turn would push
    it past `max_bytes` UTF-8 bytes; a single oversized turn still gets its own batch (inference.py:487-520)."""
    batches: List[str] = []
    cur: List[str] = []
    size = 0
    for turn in turns:
        nbytes = len(turn.encode("utf-8"))
        if cur and (len(cur) >= max_speakers or size + nbytes > max_bytes):
            batches.append("\n".join(cur))
            cur, size = [], 0
        cur.append(turn)
        size += nbytes
    if cur:
        batches.append("\n".join(cur))
    return batches


def _system_message(prompt_text, prompt_tokens) -> Message:
    """The system turn of generate_long (inference.py:565-602)."""
    if prompt_text and prompt_tokens:
        tagged = [t if _SPEAKER_TAG.search(t) else f"<|speaker:{i}|>{t}" for i, t in enumerate(prompt_text)]
        parts = [TextPart(text="convert the provided text to speech reference to the following:\n\nText:\n"),
                 TextPart(text="\n".join(tagged)),
                 TextPart(text="\n\nSpeech:\n"),
                 VQPart(codes=torch.cat([c for c in prompt_tokens], dim=1))]
    else:
        parts = [TextPart(text="convert the provided text to speech")]
    return Message(role="system", parts=parts)


def _default_generate(**kw):
    from .dual_ar import generate

    kw.pop("decode_one_token", None)
    kw.pop("audio_masks", None)
    kw.pop("audio_parts", None)
    return generate(**kw)


generate = _default_generate   # module-level so that callers (and tests) can substitute it, like the reference's


def generate_long(*, model, device: Union[str, torch.device], decode_one_token: Optional[Callable] = None,
                  text: str, num_samples: int = 1, max_new_tokens: int = 0, top_p: float = 0.9, top_k: int = 30,
                  repetition_penalty: float = 1.1, temperature: float = 1.0, compile: bool = False,
                  iterative_prompt: bool = True, chunk_length: int = 512,
                  prompt_text: Optional[Union[str, List[str]]] = None,
                  prompt_tokens: Optional[Union[torch.Tensor, List[torch.Tensor]]] = None,
                  reuse_prefix_kv: bool = True) -> Iterator[GenerateResponse]:
    """Drop-in for generate_long (inference.py:523-733); `repetition_penalty`, `compile`, `iterative_prompt` are
    accepted and unused exactly as upstream.  `reuse_prefix_kv` (extension): every chunk's prompt repeats the
    conversation so far; upstream re-prefills all of it (inference.py:620-688), here the K/V of the shared prefix stay
    in the slot and only the new columns are prefilled -- bit-identical results (tests/test_stream_gpu.py)."""
    assert 0 < top_p <= 1, "top_p must be in (0, 1]"
    assert 0 < temperature < 2, "temperature must be in (0, 2)"
    use_prompt = bool(prompt_text) and bool(prompt_tokens)
    if use_prompt and isinstance(prompt_text, str):
        prompt_text, prompt_tokens = [prompt_text], [prompt_tokens]
    if use_prompt:
        assert len(prompt_text) == len(prompt_tokens), "Prompt text and tokens must have the same length"
    if prompt_tokens:
        prompt_tokens = [c.cpu() for c in prompt_tokens]
    tokenizer = model.tokenizer
    max_length = model.config.max_seq_len
    system = _system_message(prompt_text if use_prompt else None, prompt_tokens if use_prompt else None)

    turns = split_text_by_speaker(text)
    chunks = group_turns_into_batches(turns, max_speakers=5, max_bytes=chunk_length) if turns else [text]

    for _ in range(num_samples):
        history = Conversation([system])
        for chunk in chunks:
            history.append(Message(role="user", parts=[TextPart(text=chunk)]))
            asking = history.copy()
            asking.append(Message(role="assistant", parts=[], modality="voice", add_im_end=False))
            encoded, audio_masks, audio_parts = asking.encode_for_inference(tokenizer, num_codebooks=model.config.num_codebooks)
            if encoded.size(1) > max_length - 2048:   # inference.py:658-661
                raise ValueError(f"Prompt is too long: {encoded.size(1)} > {max_length - 2048}")
            encoded = encoded.to(device=device)
            T = encoded.size(1)
            extra = {"reuse_prefix": True} if (reuse_prefix_kv and generate is _default_generate) else {}
            y = generate(model=model, prompt=encoded, max_new_tokens=max_new_tokens, audio_masks=audio_masks,
                         audio_parts=audio_parts, decode_one_token=decode_one_token, temperature=temperature,
                         top_p=top_p, top_k=top_k, **extra)
            codes = y[1:, T:-1].clone()               # drops the <|im_end|> frame (inference.py:708)
            assert (codes >= 0).all(), f"Negative code found: {codes}"
            history.append(Message(role="assistant", parts=[VQPart(codes=codes.cpu())], modality="voice"))
            yield GenerateResponse(action="sample", codes=codes, text=chunk)
        if reuse_prefix_kv and hasattr(model, "release") and generate is _default_generate:
            model.release(0)   # the next sample starts a new conversation
        yield GenerateResponse(action="next")


def init_model(checkpoint_path, device, precision=torch.bfloat16, compile: bool = False):
    """-> (model, decode_one_token) like inference.py:362-392.  `compile` is meaningless here (the frame step is
    already one hipGraph); fp16 (`precision=torch.half`, the CLI's --half) is not supported."""
    from .dual_ar import MiDualAR, decode_one_token

    if precision not in (torch.bfloat16, None):
        raise ValueError("fish_speech_amd runs the Dual-AR model in bf16 only")
    model = MiDualAR.from_pretrained(str(checkpoint_path), device=device)
    model._cache_setup_done = False
    return model.eval(), decode_one_token


def launch_thread_safe_queue(checkpoint_path, device, precision=torch.bfloat16, compile: bool = False):
    """One worker thread owns the model and serves GenerateRequests from the returned queue; `None` stops it
    (inference.py:748-799).  Errors travel back as WrappedGenerateResponse("error", exception)."""
    input_queue: queue.Queue = queue.Queue()
    ready = threading.Event()
    failure: List[BaseException] = []

    def worker():
        try:
            model, decode_one_token = init_model(checkpoint_path, device, precision, compile=compile)
            model.setup_caches(max_batch_size=1, max_seq_len=model.config.max_seq_len,
                               dtype=next(iter(model.parameters())).dtype)
        except BaseException as e:   # surface load failures to the caller instead of hanging it
            failure.append(e)
            ready.set()
            return
        ready.set()
        while True:
            item = input_queue.get()
            if item is None:
                break
            try:
                for chunk in generate_long(model=model, decode_one_token=decode_one_token, **item.request):
                    item.response_queue.put(WrappedGenerateResponse(status="success", response=chunk))
            except Exception as e:
                traceback.print_exc()
                item.response_queue.put(WrappedGenerateResponse(status="error", response=e))

    threading.Thread(target=worker, daemon=True).start()
    ready.wait()
    if failure:
        raise failure[0]
    return input_queue


def _cli():
    import click

    @click.command()
    @click.option("--text", type=str, default="<|speaker:0|>你说的对, 但是原神是一款由米哈游自主研发的开放世界手游.")
    @click.option("--prompt-text", type=str, default=None, multiple=True)
    @click.option("--prompt-tokens", type=click.Path(path_type=Path, exists=True), default=None, multiple=True)
    @click.option("--prompt-audio", type=click.Path(path_type=Path, exists=True), default=None, multiple=True)
    @click.option("--output", type=click.Path(path_type=Path), default=None)
    @click.option("--num-samples", type=int, default=1)
    @click.option("--max-new-tokens", type=int, default=0)
    @click.option("--top-p", type=float, default=0.9)
    @click.option("--top-k", type=int, default=30)
    @click.option("--temperature", type=float, default=1.0)
    @click.option("--checkpoint-path", type=click.Path(path_type=Path, exists=True), default="checkpoints/s2-pro")
    @click.option("--device", type=str, default="cuda")
    @click.option("--compile/--no-compile", default=False)
    @click.option("--seed", type=int, default=42)
    @click.option("--half/--no-half", default=False)
    @click.option("--iterative-prompt/--no-iterative-prompt", default=True)
    @click.option("--chunk-length", type=int, default=300)
    @click.option("--output-dir", type=Path, default="output")
    def main(text, prompt_text, prompt_tokens, prompt_audio, output, num_samples, max_new_tokens, top_p, top_k,
             temperature, checkpoint_path, device, compile, seed, half, iterative_prompt, chunk_length, output_dir):
        """Text -> codes_N.npy (and a wav with --output); flags of fish_speech/models/text2semantic/inference.py:802-838."""
        os.makedirs(output_dir, exist_ok=True)
        if half:
            raise click.UsageError("--half (fp16) is not supported by the MI355X path; it runs in bf16")
        if prompt_text and not prompt_audio and not prompt_tokens:
            raise ValueError("--prompt-text requires either --prompt-audio or --prompt-tokens")
        if prompt_text and prompt_tokens and len(prompt_text) != len(prompt_tokens):
            raise ValueError(f"Number of prompt text ({len(prompt_text)}) and prompt tokens ({len(prompt_tokens)}) should be the same")
        if prompt_text and prompt_audio and len(prompt_text) != len(prompt_audio):
            raise ValueError(f"Number of prompt text ({len(prompt_text)}) and prompt audio ({len(prompt_audio)}) should be the same")
        t0 = time.time()
        model, decode_one_token = init_model(checkpoint_path, device, torch.bfloat16, compile=compile)
        model.setup_caches(max_batch_size=1, max_seq_len=model.config.max_seq_len)
        print(f"Time to load model: {time.time() - t0:.02f} seconds")

        codec = None

        def get_codec():
            nonlocal codec
            if codec is None:
                from .dac import MiDAC
                # load_codec_model (inference.py:395-417): codec.to(device=device, dtype=precision), precision = bf16 --
                # the CLI runs the codec as a bf16 MODULE (parameters, activations and audio in bf16)
                codec = MiDAC.from_checkpoint(checkpoint_path / "codec.pth", device=device).to(dtype=torch.bfloat16)
            return codec

        prompt_list = None
        if prompt_audio:     # --prompt-audio takes priority over --prompt-tokens (inference.py:891-901)
            from .codec_cli import _load_wav
            prompt_list = []
            for p in prompt_audio:
                wav = _load_wav(p, get_codec().sample_rate).to(device)          # (1, 1, n)
                idx, lens = get_codec().encode(wav, torch.tensor([wav.shape[-1]], device=device))
                prompt_list.append(idx[0, :, : int(lens[0])].cpu())
        elif prompt_tokens:
            prompt_list = [torch.from_numpy(np.load(p)) for p in prompt_tokens]

        torch.manual_seed(seed)   # per-utterance sampler seeds derive from torch's seed (MiDualAR.next_seed)
        idx, codes = 0, []
        for r in generate_long(model=model, device=device, decode_one_token=decode_one_token, text=text,
                               num_samples=num_samples, max_new_tokens=max_new_tokens, top_p=top_p, top_k=top_k,
                               temperature=temperature, compile=compile, iterative_prompt=iterative_prompt,
                               chunk_length=chunk_length, prompt_text=list(prompt_text) if prompt_text else None,
                               prompt_tokens=prompt_list):
            if r.action == "sample":
                codes.append(r.codes)
                print(f"Sampled text: {r.text}")
            elif r.action == "next":
                if codes:
                    merged = torch.cat(codes, dim=1)
                    path = os.path.join(output_dir, f"codes_{idx}.npy")     # (num_codebooks, n_frames) ints
                    np.save(path, merged.cpu().numpy())
                    print(f"Saved codes to {path}")
                    if output:
                        audio = get_codec().from_indices(merged[None].to(device))[0, 0].float().cpu().numpy()
                        from scipy.io import wavfile
                        out = output if num_samples == 1 else output.with_stem(f"{output.stem}_{idx}")
                        wavfile.write(str(out), get_codec().sample_rate, audio)
                        print(f"Saved audio to {out}")
                codes = []
                idx += 1

    return main


if __name__ == "__main__":
    _cli()()

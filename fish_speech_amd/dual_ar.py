"""Host-side mirror of the reference's Dual-AR seams, driving libfishmi.so through ctypes.

Reference surface mirrored (fish_speech/models/text2semantic/):
  * ``DualARTransformer`` as used by inference.py (llama.py:660-828): ``config``, ``tokenizer``,
    ``parameters()``, ``setup_caches``, ``eval``, ``to``           -> :class:`MiDualAR`
  * ``decode_one_token_ar`` (inference.py:96-181), the callable every caller threads through
    ``generate``/``generate_long``                                  -> :func:`decode_one_token`
  * ``generate`` (inference.py:243-359)                             -> :func:`generate`
  * new capability (the reference is batch-1 only): :func:`generate_batch`.

PyTorch here is plumbing only: device buffers, streams, checkpoint reading.  Every FLOP of the
path runs in the HIP kernels of ``csrc/``; there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
import itertools
import threading
import json
import math
import os
from dataclasses import dataclass, fields
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from . import _lib
from ._lib import DualARConfigC, SamplingC, check

IM_END_TOKEN = "<|im_end|>"  # fish_speech/tokenizer.py


@dataclass
class DualARConfig:
    """Fields of DualARModelArgs (llama.py:27-193) that the inference path reads."""

    vocab_size: int
    n_layer: int
    n_head: int
    n_local_heads: int
    head_dim: int
    dim: int
    intermediate_size: int
    codebook_size: int
    num_codebooks: int
    semantic_begin_id: int
    semantic_end_id: int
    im_end_id: int
    max_seq_len: int = 2048
    rope_base: float = 10000.0
    norm_eps: float = 1e-5
    attention_qk_norm: bool = False
    scale_codebook_embeddings: bool = False
    norm_fastlayer_input: bool = False
    n_fast_layer: int = 4
    fast_dim: Optional[int] = None
    fast_n_head: Optional[int] = None
    fast_n_local_heads: Optional[int] = None
    fast_head_dim: Optional[int] = None
    fast_intermediate_size: Optional[int] = None
    fast_attention_qk_norm: Optional[bool] = None
    weight_int8: bool = False     # weight-only int8 checkpoint (tools/llama/quantize.py); set by load_state_dict's caller

    def __post_init__(self):  # defaults follow llama.py:165-193
        self.fast_dim = self.fast_dim or self.dim
        self.fast_n_head = self.fast_n_head or self.n_head
        self.fast_n_local_heads = self.fast_n_local_heads or self.n_local_heads
        self.fast_head_dim = self.fast_head_dim or self.head_dim
        self.fast_intermediate_size = self.fast_intermediate_size or self.intermediate_size
        if self.fast_attention_qk_norm is None:
            self.fast_attention_qk_norm = self.attention_qk_norm

    @classmethod
    def from_any(cls, cfg, im_end_id: Optional[int] = None) -> "DualARConfig":
        """Accept the reference's DualARModelArgs, the oracle's config, or a dict."""
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        kw = {}
        for f in fields(cls):
            v = get(f.name)
            if v is not None:
                kw[f.name] = v
        if im_end_id is not None:
            kw["im_end_id"] = im_end_id
        if "im_end_id" not in kw:
            raise ValueError("im_end_id is required (tokenizer.get_token_id('<|im_end|>'))")
        return cls(**kw)

    @classmethod
    def from_fish_qwen3_omni(cls, data: dict, im_end_id: int, semantic_begin_id: Optional[int] = None,
                             semantic_end_id: Optional[int] = None) -> "DualARConfig":
        """config.json of model_type 'fish_qwen3_omni' (llama.py:99-143), with the reference's fallbacks for
        absent keys (dataclass defaults and __post_init__, llama.py:27-72,165-193): head_dim 64, n_local_heads =
        n_head, intermediate_size = 8*dim/3 rounded up to 256, rope_base 1e4, norm_eps 1e-5, max_seq_len 2048.
        Options the MI355X path does not implement are refused instead of ignored."""
        tc, adc = data["text_config"], data["audio_decoder_config"]
        for blk, name in ((tc, "text_config"), (adc, "audio_decoder_config")):
            for opt in ("attention_qkv_bias", "attention_o_bias"):
                if blk.get(opt):
                    raise ValueError(f"{name}.{opt}=true is not supported by fish_speech_amd")
        if not tc.get("tie_word_embeddings", True):
            raise ValueError("text_config.tie_word_embeddings=false is not supported by fish_speech_amd")
        dim = tc["dim"]
        inter = tc.get("intermediate_size")
        if inter is None:
            inter = -(-int(2 * 4 * dim / 3) // 256) * 256      # find_multiple(n_hidden, 256), llama.py:67-70
        n_local = tc.get("n_local_heads", -1)
        return cls(
            vocab_size=tc["vocab_size"], n_layer=tc["n_layer"], n_head=tc["n_head"],
            n_local_heads=tc["n_head"] if n_local in (-1, None) else n_local,
            head_dim=tc.get("head_dim") or 64, dim=dim, intermediate_size=inter,
            rope_base=tc.get("rope_base", 10000), norm_eps=tc.get("norm_eps", 1e-5),
            max_seq_len=tc.get("max_seq_len", 2048), attention_qk_norm=tc.get("attention_qk_norm", False),
            semantic_begin_id=semantic_begin_id if semantic_begin_id is not None else data.get("semantic_start_token_id", 0),
            semantic_end_id=semantic_end_id if semantic_end_id is not None else data.get("semantic_end_token_id", 0),
            im_end_id=im_end_id, scale_codebook_embeddings=True, norm_fastlayer_input=True,
            codebook_size=adc["vocab_size"], num_codebooks=adc["num_codebooks"], n_fast_layer=adc["n_layer"],
            fast_dim=adc.get("dim"), fast_n_head=adc.get("n_head"), fast_n_local_heads=adc.get("n_local_heads"),
            fast_head_dim=adc.get("head_dim"), fast_intermediate_size=adc.get("intermediate_size"),
            fast_attention_qk_norm=adc.get("attention_qk_norm"),
        )

    def to_c(self) -> DualARConfigC:
        c = DualARConfigC()
        for name, _ in DualARConfigC._fields_:
            v = getattr(self, name)
            setattr(c, name, float(v) if name in ("rope_base", "norm_eps") else int(v))
        return c


def expected_state_shapes(cfg: "DualARConfig") -> Dict[str, tuple]:
    """Checkpoint tensors (name -> shape) the loader expects after the key remap."""

    def block(prefix, dim, H, KVH, D, ffn, qk):
        s = {f"{prefix}.attention.wqkv.weight": ((H + 2 * KVH) * D, dim),
             f"{prefix}.attention.wo.weight": (dim, H * D),
             f"{prefix}.feed_forward.w1.weight": (ffn, dim), f"{prefix}.feed_forward.w3.weight": (ffn, dim),
             f"{prefix}.feed_forward.w2.weight": (dim, ffn),
             f"{prefix}.ffn_norm.weight": (dim,), f"{prefix}.attention_norm.weight": (dim,)}
        if qk:
            s[f"{prefix}.attention.q_norm.weight"] = (D,)
            s[f"{prefix}.attention.k_norm.weight"] = (D,)
        return s

    out = {"embeddings.weight": (cfg.vocab_size, cfg.dim),
           "codebook_embeddings.weight": (cfg.codebook_size * cfg.num_codebooks, cfg.dim),
           "norm.weight": (cfg.dim,), "fast_embeddings.weight": (cfg.codebook_size, cfg.fast_dim),
           "fast_norm.weight": (cfg.fast_dim,), "fast_output.weight": (cfg.codebook_size, cfg.fast_dim)}
    if cfg.fast_dim != cfg.dim:      # llama.py:665-666: Linear(dim, fast_dim) with bias between the two transformers
        out["fast_project_in.weight"] = (cfg.fast_dim, cfg.dim)
        out["fast_project_in.bias"] = (cfg.fast_dim,)
    for i in range(cfg.n_layer):
        out.update(block(f"layers.{i}", cfg.dim, cfg.n_head, cfg.n_local_heads, cfg.head_dim,
                         cfg.intermediate_size, cfg.attention_qk_norm))
    for i in range(cfg.n_fast_layer):
        out.update(block(f"fast_layers.{i}", cfg.fast_dim, cfg.fast_n_head, cfg.fast_n_local_heads,
                         cfg.fast_head_dim, cfg.fast_intermediate_size, cfg.fast_attention_qk_norm))
    return out


def remap_fish_qwen3_omni_keys(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HF checkpoint names -> model names (same mapping as llama.py:229-246)."""
    if not any(k.startswith(("text_model.", "audio_decoder.")) for k in weights):
        return weights
    out = {}
    for k, v in weights.items():
        if k.startswith("text_model.model."):
            out[k[len("text_model.model."):]] = v
        elif k.startswith("audio_decoder."):
            s = k[len("audio_decoder."):]
            out[s if s.startswith("codebook_embeddings.") else "fast_" + s] = v
        else:
            out[k] = v
    return out


def _rope_table(seq_len: int, n_elem: int, base: float) -> torch.Tensor:
    """llama.py:1004-1023, built with torch on the host so the bf16 table is the reference's."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    f = torch.outer(torch.arange(seq_len), freqs)
    cis = torch.polar(torch.ones_like(f), f)
    return torch.stack([cis.real, cis.imag], dim=-1).to(torch.bfloat16).reshape(seq_len, n_elem).contiguous()


class _FakeTokenizer:
    def __init__(self, im_end_id):
        self.im_end_id = im_end_id

    def get_token_id(self, token):
        if token != IM_END_TOKEN:
            raise KeyError(token)
        return self.im_end_id


@dataclass
class ForwardResult:
    logits: torch.Tensor
    hidden_states: torch.Tensor


class MiDualAR:
    """DualARTransformer-shaped object whose compute lives in libfishmi.so."""

    def __init__(self, config, device="cuda:0", im_end_id: Optional[int] = None, tokenizer=None):
        self.lib = _lib.load()
        if tokenizer is not None and im_end_id is None:
            im_end_id = tokenizer.get_token_id(IM_END_TOKEN)
        self.config = config if isinstance(config, DualARConfig) else DualARConfig.from_any(config, im_end_id)
        self.tokenizer = tokenizer if tokenizer is not None else _FakeTokenizer(self.config.im_end_id)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.FishmiError("MiDualAR needs a GPU device (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self._c = self.config.to_c()
        need = self.lib.fmi_dualar_arena_bytes(C.byref(self._c))
        if need < 0:
            check(-1)
        # one contiguous blob: a single RCCL broadcast replicates the weights (dist.py)
        self.arena = torch.empty(need, dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        check(self.lib.fmi_dualar_create(C.byref(self._c), C.c_void_p(self.arena.data_ptr()), need, C.byref(h)))
        self._h = h
        self._cache_setup_done = False
        self.max_batch_size = -1
        self.max_seq_len = -1
        self._frame_index = 0
        self._ignore_eos = False
        self._cached_prompt: Dict[int, torch.Tensor] = {}   # slot -> prompt columns whose prefill K/V it still holds
        self.prefilled_rows = 0                             # bookkeeping for tests / reports
        self.reused_rows = 0
        self._seed_counter = itertools.count()
        # One generation call at a time per model: slots, workspaces and graphs of a handle are shared state (the
        # reference serialises LLM work through its single-worker queue, inference.py:748-799).  Entry points that
        # own a whole request (engine.StreamingTTSEngine.inference) hold this lock for its duration.  NOT re-entrant and not
        # owned by a thread: a generator that holds it across yields may be advanced, closed or finalised from another
        # thread, and two requests interleaved on ONE thread must exclude each other too.
        self.lock = threading.Lock()
        self._dtype_probe = torch.empty(0, dtype=torch.bfloat16, device=self.device)
        self.fixed_temperature = torch.tensor(0.7, device=self.device)
        self.fixed_top_p = torch.tensor(0.7, device=self.device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.fmi_dualar_destroy(h)
            self._h = None

    # ---- nn.Module-ish surface used by inference.py (277-279, 557, 377, 392)
    def parameters(self) -> Iterable[torch.Tensor]:
        yield self._dtype_probe

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- weights
    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True):
        """Row-major bf16 tensors by reference key; re-tiled inside the library."""
        state = remap_fish_qwen3_omni_keys(dict(state))
        # separate wq/wk/wv -> wqkv (Attention.load_hook, llama.py:877-882)
        for k in [k for k in state if k.endswith("attention.wq.weight")]:
            pre = k[: -len("wq.weight")]
            state[pre + "wqkv.weight"] = torch.cat([state.pop(pre + "wq.weight"), state.pop(pre + "wk.weight"),
                                                    state.pop(pre + "wv.weight")])
            if pre + "wq.scales" in state:
                state[pre + "wqkv.scales"] = torch.cat([state.pop(pre + "wq.scales"), state.pop(pre + "wk.scales"),
                                                        state.pop(pre + "wv.scales")])
        cfg = self.config
        state.setdefault("freqs_cis", _rope_table(cfg.max_seq_len, cfg.head_dim, cfg.rope_base))
        state.setdefault("fast_freqs_cis", _rope_table(cfg.num_codebooks, cfg.fast_head_dim, cfg.rope_base))
        s = self._stream()
        quantised = {k[: -len("scales")] + "weight" for k in state if k.endswith(".scales")}
        if quantised and not cfg.weight_int8:
            raise ValueError("this state dict is a weight-only int8 checkpoint: build the model with "
                             "DualARConfig(weight_int8=True) (from_pretrained does it for '*int8*' paths)")
        for name in sorted(quantised):   # int8 weight + per-row scales (WeightOnlyInt8QuantHandler, quantize.py:186-202)
            w = state[name].detach().to(self.device).contiguous()
            sc = state[name[: -len("weight")] + "scales"].detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
            if w.dtype != torch.int8:
                raise ValueError(f"{name}: expected an int8 weight next to its .scales, got {w.dtype}")
            rc = self.lib.fmi_dualar_load_tensor_int8(self._h, name.encode(), C.c_void_p(w.data_ptr()),
                                                      C.c_void_p(sc.data_ptr()), w.shape[0], w.shape[1], 1, s)
            check(rc)
            torch.cuda.current_stream(self.device).synchronize()
        for name, t in state.items():
            if name in ("causal_mask",) or name in quantised or name.endswith(".scales"):
                continue
            t = t.detach()
            if t.dtype != torch.bfloat16:
                t = t.to(torch.bfloat16)
            t = t.to(self.device).contiguous()
            rows, cols = (1, t.numel()) if t.dim() == 1 else (t.shape[0], t.numel() // t.shape[0])
            rc = self.lib.fmi_dualar_load_tensor(self._h, name.encode(), C.c_void_p(t.data_ptr()), rows, cols, 1, s)
            if rc != 0 and not strict:
                continue
            check(rc)
            torch.cuda.current_stream(self.device).synchronize()
        check(self.lib.fmi_dualar_finalize_weights(self._h, s))
        torch.cuda.current_stream(self.device).synchronize()
        return self

    def weights_ready(self, stream=None):
        """After a broadcast / copy filled the arena on a non-loading rank.  `stream` (default: torch's current stream
        on this device) is the stream that work was enqueued on; the handle's own stream is ordered after it inside the
        library, so no host synchronize is needed before the first prefill derives its tables from the arena."""
        s = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        check(self.lib.fmi_dualar_weights_ready(self._h, s))

    @classmethod
    def from_state_dict(cls, config, state, device="cuda:0", im_end_id=None) -> "MiDualAR":
        return cls(config, device=device, im_end_id=im_end_id).load_state_dict(state)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", max_length: Optional[int] = None) -> "MiDualAR":
        """Checkpoint directory loader (llama.py:480-594): config.json + safetensors / model.pth."""
        from safetensors.torch import load_file

        with open(os.path.join(path, "config.json")) as f:
            data = json.load(f)
        tokenizer = None
        try:  # the reference injects the semantic id range from its tokenizer (llama.py:499-510)
            from fish_speech.tokenizer import FishTokenizer  # type: ignore

            tokenizer = FishTokenizer.from_pretrained(path)
        except Exception:
            tokenizer = None
        if tokenizer is None:
            raise _lib.FishmiError("from_pretrained needs fish_speech.tokenizer.FishTokenizer for the id ranges")
        im_end = tokenizer.get_token_id(IM_END_TOKEN)
        if data.get("model_type") == "fish_qwen3_omni":
            cfg = DualARConfig.from_fish_qwen3_omni(data, im_end, tokenizer.semantic_begin_id,
                                                    tokenizer.semantic_end_id)
        else:
            data = dict(data, semantic_begin_id=tokenizer.semantic_begin_id,
                        semantic_end_id=tokenizer.semantic_end_id)
            cfg = DualARConfig.from_any(data, im_end)
        if max_length is not None:
            cfg.max_seq_len = max_length
        if "int8" in str(path):     # llama.py:529-534
            cfg.weight_int8 = True
        if "int4" in str(path):     # llama.py:536-544
            raise _lib.FishmiError("int4 checkpoints are not supported by fish_speech_amd")
        model = cls(cfg, device=device, tokenizer=tokenizer)
        index = os.path.join(path, "model.safetensors.index.json")
        weights: Dict[str, torch.Tensor] = {}
        if os.path.exists(index):
            with open(index) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
            for sh in shards:
                weights.update(load_file(os.path.join(path, sh), device="cpu"))
        elif os.path.exists(os.path.join(path, "model.safetensors")):
            weights = load_file(os.path.join(path, "model.safetensors"), device="cpu")
        elif os.path.exists(os.path.join(path, "model.pth")):
            weights = torch.load(os.path.join(path, "model.pth"), map_location="cpu", mmap=True, weights_only=True)
            weights = weights.get("state_dict", weights)
            if weights and next(iter(weights)).startswith("model."):
                weights = {k.replace("model.", ""): v for k, v in weights.items()}
            weights = {k: v for k, v in weights.items() if "audio_" not in k}
        else:
            raise FileNotFoundError(f"No model weights found in {path}")
        return model.load_state_dict(weights, strict=False)

    # ---- caches (llama.py:307-324,708-722)
    def setup_caches(self, max_batch_size: int, max_seq_len: int, dtype: torch.dtype = torch.bfloat16):
        if dtype != torch.bfloat16:
            raise _lib.FishmiError("only bfloat16 KV caches are implemented")
        if self.max_seq_len >= max_seq_len and self.max_batch_size >= max_batch_size:
            return
        check(self.lib.fmi_dualar_setup_caches(self._h, int(max_batch_size), int(max_seq_len)))
        self.max_batch_size, self.max_seq_len = max_batch_size, max_seq_len
        self._cache_setup_done = True

    # ---- the model-object seam (llama.py:390-466, 799-828) for callers keeping decode_one_token_ar
    def _table(self, which: int, dtype) -> torch.Tensor:
        p, r, c = C.c_void_p(), C.c_int(), C.c_int()
        check(self.lib.fmi_dualar_table_ptr(self._h, which, C.byref(p), C.byref(r), C.byref(c)))
        return _from_ptr(p.value, (r.value, c.value), dtype, self.device)

    def forward_generate(self, x: torch.Tensor, input_pos: Optional[torch.Tensor] = None, audio_masks=None,
                         audio_parts=None) -> "ForwardResult":
        """x: (1, 1+ncb, S) integer.  Returns logits (1, 1, vocab) -- finite only on the constrained rows,
        which is what survives the reference's semantic_logit_bias anyway -- and hidden_states (1, 1, fast_dim):
        what the fast transformer is handed, i.e. after `fast_project_in` when fast_dim != dim (llama.py:827)."""
        cfg = self.config
        if audio_parts is not None:   # llama.py:423-433: no audio_projector exists upstream either; it warns and goes on
            _warn_audio_parts()
        if not self._cache_setup_done:
            self.setup_caches(1, cfg.max_seq_len)
        ncb1 = cfg.num_codebooks + 1
        xs = x.reshape(ncb1, -1).t().to(device=self.device, dtype=torch.int32).contiguous()
        pos0 = 0 if input_pos is None else int(input_pos.reshape(-1)[0].item())
        self._check_tokens(xs)
        self._cached_prompt.pop(0, None)           # slot 0's K/V are rewritten: a retained prefix is stale now
        ids = self._table(1, torch.int32).view(-1).long()
        live = torch.empty(ids.numel(), dtype=torch.bfloat16, device=self.device)
        hidden = torch.empty(cfg.fast_dim, dtype=torch.bfloat16, device=self.device)
        check(self.lib.fmi_dualar_forward_slow(self._h, 0, C.c_void_p(xs.data_ptr()), int(xs.shape[0]), pos0,
                                               C.c_void_p(live.data_ptr()), C.c_void_p(hidden.data_ptr()), self._stream()))
        logits = torch.full((1, 1, cfg.vocab_size), float("-inf"), dtype=torch.bfloat16, device=self.device)
        logits[0, 0, ids] = live
        self._keep = xs
        return ForwardResult(logits=logits, hidden_states=hidden.view(1, 1, -1))

    def forward_generate_fast(self, x: torch.Tensor, input_pos: torch.Tensor) -> torch.Tensor:
        cfg = self.config
        hid = x.reshape(-1).to(device=self.device, dtype=torch.bfloat16).contiguous()
        out = torch.empty(cfg.codebook_size, dtype=torch.bfloat16, device=self.device)
        check(self.lib.fmi_dualar_forward_fast(self._h, 0, C.c_void_p(hid.data_ptr()), int(input_pos.reshape(-1)[0].item()),
                                               C.c_void_p(out.data_ptr()), self._stream()))
        self._keep = hid
        return out.view(1, 1, -1)

    def fast_embeddings(self, idx: torch.Tensor) -> torch.Tensor:
        return self._table(0, torch.bfloat16)[idx.to(self.device).long()]

    # ---- batched API (new capability)
    def _sampling(self, temperature, top_p, top_k, seed, use_ras=True) -> SamplingC:
        return SamplingC(float(temperature), float(top_p), int(top_k), int(seed) & 0xFFFFFFFF, int(bool(use_ras)))

    def next_seed(self) -> int:
        return (torch.initial_seed() + 0x9E37 * next(self._seed_counter)) & 0xFFFFFFFF

    def _check_tokens(self, rows: torch.Tensor):
        """rows: (S, 1+ncb) integer.  The embedding kernel gathers E[row 0] and CB[code + i*codebook_size] without
        bounds checks; the reference's nn.Embedding raises IndexError on an out-of-range id, so do the same here
        (row 0 in [0, vocab), code rows in [0, codebook_size) -- they index the shared codebook table, llama.py:403)."""
        cfg = self.config
        t0, codes = rows[:, 0], rows[:, 1:]
        if rows.numel() and (int(t0.min()) < 0 or int(t0.max()) >= cfg.vocab_size):
            raise IndexError(f"token id out of range [0, {cfg.vocab_size})")
        if codes.numel() and (int(codes.min()) < 0 or int(codes.max()) >= cfg.codebook_size):
            raise IndexError(f"codebook index out of range [0, {cfg.codebook_size})")

    def _reusable_prefix(self, slot: int, prompt: torch.Tensor) -> int:
        """Number of leading columns of `prompt` whose K/V the slot still holds from an earlier PREFILL of the same
        columns (0 if none).  At least one column is always left to run: it produces the frame's logits."""
        old = self._cached_prompt.get(slot)
        if old is None:
            return 0
        new = prompt.detach().to("cpu", torch.int64)
        # prompts of up to 16 rows run through the decode GEMV, longer ones through the tiled GEMM: K/V written by one
        # are not bit-identical to what the other would write, so a prefix is only reused within the same class
        if (old.shape[1] > 16) != (new.shape[1] > 16):
            return 0
        m = min(old.shape[1], new.shape[1] - 1)
        if m <= 0:
            return 0
        same = (old[:, :m] == new[:, :m]).all(dim=0)
        bad = (~same).nonzero()
        return int(bad[0]) if len(bad) else m

    def prefill(self, slots: Sequence[int], prompts: Sequence[torch.Tensor], max_new_tokens: Sequence[int],
                sampling: Sequence[SamplingC], reuse_prefix: bool = False):
        """prompts[i]: (1+ncb, T_i) integer tensor (the reference's prompt layout).  With `reuse_prefix` the slot's
        pages are kept between calls (do not release it) and only the columns after the longest prefix the cache
        already holds are run (fmi_dualar_prefill_resume): bit-identical to running the whole prompt."""
        n = len(slots)
        pos0 = [0] * n
        # prefix-KV reuse (generate_long's chunks): skip what the slot's cache already holds.  Single-prompt calls only:
        # the C side picks the GEMV or the tiled GEMM from the TOTAL row count of a call, so in a batch of short
        # prompts the kernel class of a resumed suffix could differ from what a full prefill of the batch would use
        # and the documented bit-identity with re-prefill would not hold.
        reuse_prefix = reuse_prefix and n == 1
        if reuse_prefix:
            pos0 = [self._reusable_prefix(int(s), p) for s, p in zip(slots, prompts)]
        for s, p0 in zip(slots, pos0):
            if p0 == 0 and self._cached_prompt.pop(int(s), None) is not None:
                self.release(int(s))          # the retained pages belong to another conversation
        toks = torch.cat([p[:, p0:].to(self.device).t().to(torch.int32) for p, p0 in zip(prompts, pos0)], dim=0).contiguous()
        self._check_tokens(toks)
        lens = (C.c_int32 * n)(*[int(p.shape[1]) - p0 for p, p0 in zip(prompts, pos0)])
        sl = (C.c_int32 * n)(*[int(s) for s in slots])
        mn = (C.c_int32 * n)(*[int(m) for m in max_new_tokens])
        sp = (SamplingC * n)(*sampling)
        if any(pos0):
            p0c = (C.c_int32 * n)(*pos0)
            check(self.lib.fmi_dualar_prefill_resume(self._h, n, sl, C.c_void_p(toks.data_ptr()), lens, p0c, mn, sp,
                                                     self._stream()))
        else:
            check(self.lib.fmi_dualar_prefill(self._h, n, sl, C.c_void_p(toks.data_ptr()), lens, mn, sp, self._stream()))
        self.prefilled_rows += sum(int(p.shape[1]) - p0 for p, p0 in zip(prompts, pos0))
        self.reused_rows += sum(pos0)
        if reuse_prefix:   # positions [0, T) now hold prefill-written K/V of exactly these columns
            for s, p in zip(slots, prompts):
                self._cached_prompt[int(s)] = p.detach().to("cpu", torch.int64).clone()
        self._keep = toks

    def decode(self, slots: Sequence[int], n_frames: int):
        n = len(slots)
        sl = (C.c_int32 * n)(*[int(s) for s in slots])
        check(self.lib.fmi_dualar_decode(self._h, n, sl, int(n_frames), self._stream()))

    def poll_done(self, slots: Sequence[int]) -> List[int]:
        n = len(slots)
        sl = (C.c_int32 * n)(*[int(s) for s in slots])
        out = (C.c_int32 * n)()
        check(self.lib.fmi_dualar_poll_done(self._h, n, sl, out, self._stream()))
        return list(out)

    def read(self, slot: int):
        """-> (frames (n, 1+ncb) int32 CPU tensor, done flag)."""
        cap = self.max_seq_len
        ncb1 = self.config.num_codebooks + 1
        buf = (C.c_int32 * (cap * ncb1))()
        n, done = C.c_int(), C.c_int()
        check(self.lib.fmi_dualar_read(self._h, int(slot), buf, cap, C.byref(n), C.byref(done), self._stream()))
        t = torch.frombuffer(buf, dtype=torch.int32)[: n.value * ncb1].reshape(n.value, ncb1).clone()
        return t, done.value

    def frames_device(self, n_slots: int, n_frames: int) -> torch.Tensor:
        """Device view of the frames generated so far: (n_slots, n_frames, 1+ncb) int32 (slots 0..n-1)."""
        p, mf = C.c_void_p(), C.c_int()
        check(self.lib.fmi_dualar_out_ptr(self._h, C.byref(p), C.byref(mf)))
        ncb1 = self.config.num_codebooks + 1
        full = _from_ptr(p.value, (self.max_batch_size, mf.value, ncb1), torch.int32, self.device)
        check(self.lib.fmi_dualar_synchronize(self._h))   # decode() leaves the caller's stream unordered (fishmi.h)
        return full[:n_slots, :n_frames]

    def synchronize(self):
        """host wait for everything enqueued on this model's stream"""
        check(self.lib.fmi_dualar_synchronize(self._h))

    def wait_stream(self):
        """order torch's current stream after the frames of the last decode() call (no host wait)"""
        check(self.lib.fmi_dualar_wait(self._h, self._stream()))

    def release(self, slot: int):
        self._cached_prompt.pop(int(slot), None)
        check(self.lib.fmi_dualar_release(self._h, int(slot)))

    def last_decode_stats(self):
        ms, nl = C.c_float(), C.c_int()
        check(self.lib.fmi_dualar_last_decode_stats(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def derived_info(self) -> Dict[str, int]:
        """What this handle derived from its arena locally (dist.py: not part of the broadcast) and how many tensors it
        was given through load_state_dict (0 on a rank fed by the broadcast + weights_ready alone)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.fmi_dualar_derived_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"row_copies": a.value, "table_rows": b.value, "loaded_tensors": c.value}

    def set_graph(self, enable: bool):
        check(self.lib.fmi_dualar_set_graph(self._h, int(enable)))

    def set_attn_impl(self, impl: int):
        """Prefill attention kernel: 1 = MFMA flash attention (default), 0 = VALU kernel (A/B parity runs)."""
        check(self.lib.fmi_dualar_set_attn_impl(self._h, int(impl)))

    def set_stream_priority(self, priority: int):
        """Dispatch priority of the model's private stream: -1 highest, 0 default, 1 lowest (re-creates the stream)."""
        check(self.lib.fmi_dualar_set_stream_priority(self._h, int(priority)))

    def set_fast_merge(self, enable: bool):
        """Fast positions 0 and 1 of a frame in one pass over the fast weights (default) or in two (A/B parity runs)."""
        check(self.lib.fmi_dualar_set_fast_merge(self._h, int(bool(enable))))

    def set_attn_long_threshold(self, threshold: int):
        """Decode attention: rows at or beyond this position use the MFMA kernel + merge, the others the fused VALU
        kernel (default 1024; 0 = VALU for every row).  Call after setup_caches."""
        check(self.lib.fmi_dualar_set_attn_long_threshold(self._h, int(threshold)))

    def set_ignore_eos(self, enable: bool):
        """Keep generating past <|im_end|> (fixed-length synthetic benchmarks)."""
        check(self.lib.fmi_dualar_set_ignore_eos(self._h, int(enable)))
        self._ignore_eos = bool(enable)

    # ---- parity taps
    def debug_taps(self, B: int = 1):
        """(slow live-row logits (B, n_live) bf16, live vocab ids, the hidden rows the fast transformer is handed
        (B, fast_dim), last fast logits)."""
        p_log, p_ids, p_hid, p_fl = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n_live, ld = C.c_int(), C.c_int()
        check(self.lib.fmi_dualar_debug_ptrs(self._h, C.byref(p_log), C.byref(n_live), C.byref(ld), C.byref(p_ids),
                                             C.byref(p_hid), C.byref(p_fl)))
        torch.cuda.synchronize(self.device)
        cfg = self.config
        logits = _from_ptr(p_log.value, (B, ld.value), torch.bfloat16, self.device)[:, : n_live.value].clone()
        ids = _from_ptr(p_ids.value, (n_live.value,), torch.int32, self.device).clone()
        hidden = _from_ptr(p_hid.value, (B, cfg.fast_dim), torch.bfloat16, self.device).clone()   # projected when fast_dim != dim
        fl = _from_ptr(p_fl.value, (B, cfg.codebook_size), torch.bfloat16, self.device).clone()
        return logits, ids, hidden, fl

    def set_trace(self, enable: bool):
        p = C.c_void_p()
        check(self.lib.fmi_dualar_set_trace(self._h, int(enable), C.byref(p)))
        self._trace_ptr = p.value

    def fast_chain_forced(self, hidden: torch.Tensor, forced: torch.Tensor, slots, table: bool = True) -> torch.Tensor:
        """Test seam (fishmi.h: fmi_dualar_fast_chain_forced): the fast chain of one frame for `slots` from the normed
        hidden rows `hidden` (B, dim), every draw replaced by `forced` (B, 1 + num_codebooks), on the frame loop's own
        path.  Returns the fast logits (B, num_codebooks, codebook_size); row 0 of each is not computed."""
        cfg = self.config
        B = len(slots)
        hid = hidden.reshape(B, -1).to(device=self.device, dtype=torch.bfloat16).contiguous()
        frc = forced.reshape(B, cfg.num_codebooks + 1).to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.zeros(B, cfg.num_codebooks, cfg.codebook_size, dtype=torch.bfloat16, device=self.device)
        sl = (C.c_int32 * B)(*[int(x) for x in slots])
        check(self.lib.fmi_dualar_fast_chain_forced(self._h, B, sl, C.c_void_p(hid.data_ptr()), C.c_void_p(frc.data_ptr()),
                                                    int(bool(table)), C.c_void_p(out.data_ptr()), self._stream()))
        self._keep = (hid, frc)
        return out

    def fast_trace(self, B: int = 1) -> torch.Tensor:
        cfg = self.config
        torch.cuda.synchronize(self.device)
        return _from_ptr(self._trace_ptr, (B, cfg.num_codebooks, cfg.codebook_size), torch.bfloat16,
                         self.device).clone()

    def step(self, x: torch.Tensor, pos0: int, sampling: SamplingC, previous_tokens: Optional[torch.Tensor],
             frame_index: int, slot: int = 0) -> torch.Tensor:
        """One frame for one slot = the decode_one_token seam.  x: (S, 1+ncb) int32 on device."""
        ncb1 = self.config.num_codebooks + 1
        self._check_tokens(x)
        self._cached_prompt.pop(int(slot), None)   # this call rewrites the slot's K/V: a retained prefix is stale now
        out = torch.empty(ncb1, dtype=torch.int32, device=self.device)
        prev = None
        if previous_tokens is not None:
            prev = previous_tokens.to(device=self.device, dtype=torch.int32).contiguous()
        check(self.lib.fmi_dualar_step(self._h, slot, C.c_void_p(x.data_ptr()), int(x.shape[0]), int(pos0),
                                       C.byref(sampling), C.c_void_p(prev.data_ptr()) if prev is not None else None,
                                       int(frame_index), C.c_void_p(out.data_ptr()), self._stream()))
        self._keep = (x, prev)
        return out


class _CudaArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": tuple(shape), "typestr": typestr,
                                         "version": 2, "strides": None}


def _from_ptr(ptr: int, shape, dtype, device) -> torch.Tensor:
    """View library-owned device memory as a tensor (tests / debug only)."""
    n = int(math.prod(shape))
    esize = torch.empty(0, dtype=dtype).element_size()
    raw = torch.as_tensor(_CudaArray(ptr, (n * esize,), "|u1"), device=device)
    return raw.view(dtype).reshape(shape)


# --------------------------------------------------------------------------- reference-shaped callables


def _require_default_bias(model: MiDualAR, bias: torch.Tensor):
    """The library scores only the rows generate() leaves finite (inference.py:310-320: 0 on the semantic ids and
    <|im_end|>, -inf elsewhere).  A caller-supplied bias is accepted only if it IS that mask; anything else would be
    silently ignored, so it is refused."""
    key = (bias.data_ptr(), tuple(bias.shape))
    if getattr(model, "_bias_ok", None) == key:
        return
    cfg = model.config
    b = bias.detach().reshape(-1).float().cpu()
    want = torch.full((cfg.vocab_size,), float("-inf"))
    want[cfg.semantic_begin_id: cfg.semantic_end_id + 1] = 0.0
    want[cfg.im_end_id] = 0.0
    if b.numel() != cfg.vocab_size or not torch.equal(b, want):
        raise NotImplementedError("fish_speech_amd.decode_one_token only supports the semantic_logit_bias that "
                                  "generate() builds (0 on semantic ids and <|im_end|>, -inf elsewhere)")
    model._bias_ok = key


def _warn_audio_parts() -> None:
    """The reference's own words (llama.py:433): S2 has no audio_projector, the argument is dead upstream too."""
    import logging

    logging.getLogger("fish_speech_amd").warning("audio_parts provided but model has no audio_projector")


def decode_one_token(model: MiDualAR, x: torch.Tensor, input_pos: torch.Tensor, temperature, top_p, top_k: int,
                     semantic_logit_bias=None, audio_masks=None, audio_parts=None,
                     previous_tokens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Drop-in for decode_one_token_ar (inference.py:96-181): same arguments, same (1+ncb, 1) int
    result.  ``semantic_logit_bias`` is implied by the config (only semantic ids + <|im_end|> are ever
    scored: the algorithmic minimum of the reference's -inf bias); passing that very mask is accepted, any other
    bias raises NotImplementedError instead of being ignored; audio_* are dead for S2."""
    ncb1 = model.config.num_codebooks + 1
    if audio_parts is not None:
        _warn_audio_parts()
    if semantic_logit_bias is not None:
        _require_default_bias(model, semantic_logit_bias)
    xs = x.reshape(ncb1, -1).t().to(device=model.device, dtype=torch.int32).contiguous()
    S = xs.shape[0]
    pos0 = int(input_pos.reshape(-1)[0].item())
    if S > 1 or previous_tokens is None:
        model._frame_index = 0
        model._call_seed = model.next_seed()
    else:
        model._frame_index += 1
    sp = model._sampling(float(temperature), float(top_p), top_k, getattr(model, "_call_seed", 0),
                         previous_tokens is not None)
    out = model.step(xs, pos0, sp, previous_tokens, model._frame_index)
    return out.view(ncb1, 1)


@torch.no_grad()
def generate(*, model: MiDualAR, prompt: torch.Tensor, max_new_tokens: int, audio_masks=None, audio_parts=None,
             decode_one_token=None, num_samples: int = 1, poll_every: int = 16, seed: Optional[int] = None,
             stop_on_im_end: bool = True, reuse_prefix: bool = False, **sampling_kwargs) -> torch.Tensor:
    """Drop-in for generate (inference.py:243-359) on one utterance, without the per-frame host sync:
    frames advance by hipGraph replay and <|im_end|> is polled every ``poll_every`` frames.  ``reuse_prefix``: keep
    the slot's K/V afterwards and, next time, run only the prompt columns beyond the longest prefix it shares with
    this prompt (generate_long's chunks repeat the whole conversation so far) -- results are bit-identical."""
    if audio_parts is not None:
        _warn_audio_parts()
    return generate_batch(model=model, prompts=[prompt], max_new_tokens=max_new_tokens, poll_every=poll_every,
                          seeds=None if seed is None else [seed], stop_on_im_end=stop_on_im_end,
                          reuse_prefix=reuse_prefix, **sampling_kwargs)[0]


@torch.no_grad()
def generate_batch(*, model: MiDualAR, prompts: Sequence[torch.Tensor], max_new_tokens: int, poll_every: int = 16,
                   seeds: Optional[Sequence[int]] = None, stop_on_im_end: bool = True, temperature: float = 1.0,
                   top_p: float = 0.9, top_k: int = 30, use_ras: bool = True, reuse_prefix: bool = False
                   ) -> List[torch.Tensor]:
    """Batch of utterances through prefill + graph-replayed decode.  Each result equals what the
    batch-1 path yields for that utterance (kernels are batch-invariant).  Returns, per utterance,
    (1+ncb, T_i + n_i) like the reference's ``generate``."""
    cfg = model.config
    n = len(prompts)
    for p in prompts:
        if p.size(1) >= cfg.max_seq_len:  # inference.py:263-266
            raise ValueError(f"Input sequence length {p.size(1)} exceeds max_seq_len {cfg.max_seq_len}")
    if not model._cache_setup_done:
        model.setup_caches(max_batch_size=max(n, 1), max_seq_len=cfg.max_seq_len)
    if n > model.max_batch_size:
        raise ValueError(f"batch {n} exceeds max_batch_size {model.max_batch_size}")
    mn = []
    for p in prompts:
        T = p.size(1)
        m = max_new_tokens if max_new_tokens else cfg.max_seq_len - T
        mn.append(min(m, cfg.max_seq_len - T))
    slots = list(range(n))
    seeds = list(seeds) if seeds is not None else [model.next_seed() for _ in range(n)]
    samp = [model._sampling(temperature, top_p, top_k, seeds[i], use_ras) for i in range(n)]
    model.prefill(slots, prompts, mn, samp, reuse_prefix=reuse_prefix)
    remaining = max(mn) - 1
    while remaining > 0:
        step = min(poll_every, remaining)
        model.decode(slots, step)
        remaining -= step
        if stop_on_im_end and all(model.poll_done(slots)):
            break
    outs = []
    for i, p in enumerate(prompts):
        frames, done = model.read(i)
        seq = torch.cat([p.to("cpu", torch.int64), frames.t().to(torch.int64)], dim=1)
        outs.append(seq.to(p.dtype) if p.dtype in (torch.int32, torch.int64) else seq)
        if not reuse_prefix:
            model.release(i)
    return outs


@torch.no_grad()
def generate_batch_device(*, model: MiDualAR, prompts: Sequence[torch.Tensor], max_new_tokens: int,
                          seeds: Optional[Sequence[int]] = None, temperature: float = 1.0, top_p: float = 0.9,
                          top_k: int = 30, use_ras: bool = True, wait: bool = True):
    """Fixed-length batch generation whose result stays on the device: returns the codebook rows
    (B, num_codebooks, max_new_tokens) int64, ready for ``MiDAC.from_indices`` -- no host round trip
    between the Dual-AR loop and the codec.  Every slot runs exactly ``max_new_tokens`` frames
    (use ``model.set_ignore_eos(True)``), so this is the serving/benchmark path for known lengths.
    ``wait=False``: returns as soon as the prefill and the frames are ENQUEUED on the model's stream (no host wait);
    pass the returned token to :func:`finish_batch_device` for the codes."""
    cfg = model.config
    n = len(prompts)
    if not model._cache_setup_done:
        model.setup_caches(max_batch_size=n, max_seq_len=cfg.max_seq_len)
    if max_new_tokens < 1:
        raise ValueError("generate_batch_device needs an explicit max_new_tokens >= 1")
    for p in prompts:   # every slot must really produce max_new_tokens frames, or the returned block holds stale rows
        if p.size(1) + max_new_tokens > model.max_seq_len:
            raise ValueError(f"prompt of {p.size(1)} tokens + {max_new_tokens} new exceeds max_seq_len {model.max_seq_len}")
    if not model._ignore_eos:
        raise ValueError("generate_batch_device returns a fixed-length block: call model.set_ignore_eos(True) first "
                         "(or use generate_batch, which returns per-utterance lengths)")
    slots = list(range(n))
    seeds = list(seeds) if seeds is not None else [model.next_seed() for _ in range(n)]
    samp = [model._sampling(temperature, top_p, top_k, seeds[i], use_ras) for i in range(n)]
    model.prefill(slots, prompts, [max_new_tokens] * n, samp)
    if max_new_tokens > 1:
        model.decode(slots, max_new_tokens - 1)
    if not wait:
        return (n, max_new_tokens)
    return finish_batch_device(model, (n, max_new_tokens))


def finish_batch_device(model: MiDualAR, pending) -> torch.Tensor:
    """Second half of ``generate_batch_device(..., wait=False)``: waits for the frames (host), copies the codebook rows
    out of the library's buffer and releases the slots.  Between the two halves the host is free -- bench.py and
    serving enqueue the PREVIOUS batch's codec decode there, so that the MFMA-bound codec runs beside the HBM-bound
    frame loop instead of after it."""
    n, max_new_tokens = pending
    out = model.frames_device(n, max_new_tokens)          # (B, frames, 1+ncb) int32 view
    codes = out[:, :, 1:].permute(0, 2, 1).to(torch.int64).contiguous()
    for i in range(n):
        model.release(i)
    return codes

"""Host-side mirror of the reference's codec seam, driving libfishmi.so through ctypes.

Reference surface mirrored (fish_speech/models/dac/modded_dac.py): ``DAC.encode`` (874-923),
``DAC.from_indices`` (925-927), attributes ``sample_rate``, ``frame_length``, ``device``,
``parameters()`` -- what VQManager, the codec CLI (dac/inference.py:90,112) and the text2semantic CLI
(inference.py:435,443) touch.  The convolutions, transformers and quantizer all run in the HIP
kernels of csrc/dac_kernels.hip; torch is used for buffers and for folding weight-norm at load time
(``torch._weight_norm``, the very function the reference's parametrization evaluates).
"""
from __future__ import annotations

import ctypes as C
import itertools
import math
import threading
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DacConfigC, check


@dataclass
class DacConfig:
    """fish_speech/configs/modded_dac_vq.yaml."""

    encoder_dim: int = 64
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)
    decoder_dim: int = 1536
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    n_codebooks: int = 9
    codebook_size: int = 1024
    semantic_codebook_size: int = 4096
    codebook_dim: int = 8
    downsample: Tuple[int, ...] = (2, 2)
    tf_layers: int = 8
    tf_ffn_mult: int = 3
    tf_window: int = 128
    enc_tf_layers: int = 4
    enc_tf_window: int = 512
    sample_rate: int = 44100

    @classmethod
    def from_any(cls, cfg) -> "DacConfig":
        kw = {f: getattr(cfg, f) for f in cls.__dataclass_fields__ if hasattr(cfg, f)}
        return cls(**kw)

    @classmethod
    def from_state_dict(cls, folded: Dict[str, torch.Tensor], **overrides) -> "DacConfig":
        """Architecture read off the tensor shapes of a (weight-norm-folded) codec checkpoint; for the released
        codec.pth this reproduces modded_dac_vq.yaml.  The attention windows are not visible in the weights: they
        keep the yaml's values unless overridden."""
        def count(prefix, suffix):
            n = 0
            while f"{prefix}{n}{suffix}" in folded:
                n += 1
            return n

        enc = folded["encoder.block.0.conv.weight"].shape[0]
        enc_rates = tuple(folded[f"encoder.block.{i + 1}.block.4.conv.weight"].shape[2] // 2 for i in range(4))
        dec = folded["decoder.model.0.conv.weight"].shape[0]
        dec_rates = tuple(folded[f"decoder.model.{i + 1}.block.1.conv.weight"].shape[2] // 2 for i in range(4))
        L = folded["decoder.model.0.conv.weight"].shape[1]
        cb = folded["quantizer.quantizer.quantizers.0.codebook.weight"]
        kw = dict(encoder_dim=enc, encoder_rates=enc_rates, decoder_dim=dec, decoder_rates=dec_rates,
                  n_codebooks=count("quantizer.quantizer.quantizers.", ".codebook.weight"), codebook_size=cb.shape[0],
                  codebook_dim=cb.shape[1],
                  semantic_codebook_size=folded["quantizer.semantic_quantizer.quantizers.0.codebook.weight"].shape[0],
                  downsample=tuple(folded[f"quantizer.downsample.{i}.0.conv.weight"].shape[2]
                                   for i in range(count("quantizer.downsample.", ".0.conv.weight"))),
                  tf_layers=count("quantizer.pre_module.layers.", ".attention.wqkv.weight"),
                  enc_tf_layers=count("encoder.block.4.block.5.layers.", ".attention.wqkv.weight"))
        if kw["tf_layers"]:
            kw["tf_ffn_mult"] = folded["quantizer.pre_module.layers.0.feed_forward.w1.weight"].shape[0] // L
        kw.update(overrides)
        return cls(**kw)

    @property
    def latent_dim(self) -> int:
        return self.encoder_dim * 2 ** len(self.encoder_rates)

    @property
    def frame_length(self) -> int:
        return int(math.prod(self.encoder_rates)) * int(math.prod(self.downsample))

    def to_c(self) -> DacConfigC:
        c = DacConfigC()
        c.encoder_dim, c.decoder_dim, c.latent_dim = self.encoder_dim, self.decoder_dim, self.latent_dim
        c.encoder_rates = (C.c_int32 * 4)(*self.encoder_rates)
        c.decoder_rates = (C.c_int32 * 4)(*self.decoder_rates)
        c.n_codebooks, c.codebook_size = self.n_codebooks, self.codebook_size
        c.semantic_codebook_size, c.codebook_dim = self.semantic_codebook_size, self.codebook_dim
        c.downsample = (C.c_int32 * 2)(*self.downsample)
        c.tf_layers, c.tf_heads = self.tf_layers, self.latent_dim // 64
        c.tf_ffn, c.tf_window = self.latent_dim * self.tf_ffn_mult, self.tf_window
        c.enc_tf_layers, c.enc_tf_window, c.sample_rate = self.enc_tf_layers, self.enc_tf_window, self.sample_rate
        return c


def fold_weight_norm(state: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    """codec.pth keys -> plain weights: w = g * v / ||v|| over all dims but 0, for both weight-norm
    spellings (parametrizations.weight.original0/1, modded_dac.py:554-556; weight_g/weight_v of the
    third-party quantizer projections).  SURVEY.md A.6.

    dtype = torch.bfloat16: the parameters of `codec.to(dtype=torch.bfloat16)` (how the text2semantic CLI holds the
    codec, text2semantic/inference.py:416): g, v and every other floating tensor are rounded to bf16 and the weight-norm
    product is evaluated in bf16 by the same torch function the parametrization calls."""
    def c(t):
        return t.to(dtype) if t.is_floating_point() else t

    out = {}
    for k, v in state.items():
        if k.endswith("parametrizations.weight.original1"):
            base = k[: -len("parametrizations.weight.original1")]
            out[base + "weight"] = torch._weight_norm(c(v.float()), c(state[base + "parametrizations.weight.original0"].float()), 0)
        elif k.endswith("weight_v"):
            base = k[: -len("weight_v")]
            out[base + "weight"] = torch._weight_norm(c(v.float()), c(state[base + "weight_g"].float()), 0)
        elif k.endswith("parametrizations.weight.original0") or k.endswith("weight_g"):
            continue
        elif "causal_mask" in k or "freqs_cis" in k:
            continue
        else:
            out[k] = c(v)
    return out


def expected_state_shapes(cfg: DacConfig) -> Dict[str, tuple]:
    """Folded codec tensors (name -> shape) the loader expects: codec.pth keys after weight-norm folding
    (SURVEY.md A.6)."""
    s: Dict[str, tuple] = {}

    def conv(p, cout, cin, k, tr=False):
        s[p + ".conv.weight"] = (cin, cout, k) if tr else (cout, cin, k)
        s[p + ".conv.bias"] = (cout,)

    def res_unit(p, d):
        s[p + ".block.0.alpha"] = (1, d, 1)
        conv(p + ".block.1", d, d, 7)
        s[p + ".block.2.alpha"] = (1, d, 1)
        conv(p + ".block.3", d, d, 1)

    def tf(p, d, n, ffn):
        for i in range(n):
            l = f"{p}.layers.{i}"
            s[l + ".attention.wqkv.weight"] = (3 * d, d)
            s[l + ".attention.wo.weight"] = (d, d)
            s[l + ".feed_forward.w1.weight"] = (ffn, d)
            s[l + ".feed_forward.w3.weight"] = (ffn, d)
            s[l + ".feed_forward.w2.weight"] = (d, ffn)
            for n2 in ("ffn_norm.weight", "attention_norm.weight", "attention_layer_scale.gamma", "ffn_layer_scale.gamma"):
                s[f"{l}.{n2}"] = (d,)
        s[p + ".norm.weight"] = (d,)

    d = cfg.encoder_dim
    conv("encoder.block.0", d, 1, 7)
    for bi, stride in enumerate(cfg.encoder_rates):
        d *= 2
        p = f"encoder.block.{bi + 1}"
        for r in range(3):
            res_unit(f"{p}.block.{r}", d // 2)
        s[f"{p}.block.3.alpha"] = (1, d // 2, 1)
        conv(f"{p}.block.4", d, d // 2, 2 * stride)
        if bi == len(cfg.encoder_rates) - 1 and cfg.enc_tf_layers:
            tf(f"{p}.block.5", d, cfg.enc_tf_layers, 3 * d)
    s["encoder.block.5.alpha"] = (1, d, 1)
    L = cfg.latent_dim
    conv("encoder.block.6", L, d, 3)
    for name, n, size in (("semantic_quantizer", 1, cfg.semantic_codebook_size), ("quantizer", cfg.n_codebooks, cfg.codebook_size)):
        for i in range(n):
            p = f"quantizer.{name}.quantizers.{i}"
            s[p + ".in_proj.weight"] = (cfg.codebook_dim, L, 1)
            s[p + ".in_proj.bias"] = (cfg.codebook_dim,)
            s[p + ".out_proj.weight"] = (L, cfg.codebook_dim, 1)
            s[p + ".out_proj.bias"] = (L,)
            s[p + ".codebook.weight"] = (size, cfg.codebook_dim)
    for name in ("downsample", "upsample"):
        for i in range(len(cfg.downsample)):
            p = f"quantizer.{name}.{i}"
            fac = cfg.downsample[i] if name == "downsample" else list(reversed(cfg.downsample))[i]
            s[p + ".0.conv.weight"] = (L, L, fac)
            s[p + ".0.conv.bias"] = (L,)
            s[p + ".1.gamma"] = (L,)
            s[p + ".1.dwconv.conv.weight"] = (L, 1, 7)
            s[p + ".1.dwconv.conv.bias"] = (L,)
            s[p + ".1.norm.weight"] = (L,)
            s[p + ".1.norm.bias"] = (L,)
            s[p + ".1.pwconv1.weight"] = (4 * L, L)
            s[p + ".1.pwconv1.bias"] = (4 * L,)
            s[p + ".1.pwconv2.weight"] = (L, 4 * L)
            s[p + ".1.pwconv2.bias"] = (L,)
    for name in ("pre_module", "post_module"):
        tf(f"quantizer.{name}", L, cfg.tf_layers, L * cfg.tf_ffn_mult)
    conv("decoder.model.0", cfg.decoder_dim, L, 7)
    for i, stride in enumerate(cfg.decoder_rates):
        cin, cout = cfg.decoder_dim // 2 ** i, cfg.decoder_dim // 2 ** (i + 1)
        p = f"decoder.model.{i + 1}"
        s[p + ".block.0.alpha"] = (1, cin, 1)
        conv(p + ".block.1", cout, cin, 2 * stride, tr=True)
        for r in range(3):
            res_unit(f"{p}.block.{2 + r}", cout)
    cl = cfg.decoder_dim // 2 ** len(cfg.decoder_rates)
    s["decoder.model.5.alpha"] = (1, cl, 1)
    conv("decoder.model.6", 1, cl, 7)
    return s


def _rope_table(n_pos: int, n_elem: int = 64, base: float = 10000.0) -> torch.Tensor:
    """modded_dac.py:442-452 (bf16 by default), built with torch so the table is the reference's."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    f = torch.outer(torch.arange(n_pos), freqs)
    cis = torch.polar(torch.ones_like(f), f)
    return torch.stack([cis.real, cis.imag], dim=-1).to(torch.bfloat16).reshape(n_pos, n_elem).contiguous()


class MiDAC:
    """DAC-shaped codec object (encode / from_indices) whose compute lives in libfishmi.so."""

    def __init__(self, config=None, device="cuda:0", check_overflow: bool = False):
        """check_overflow: after every decode-side call in the default fp16-split arithmetic, read the library's sticky
        overflow flag (one host wait per call) and, if an operand left the fp16 range (|x| >= 65504 -- the reference's
        fp32 / bf16 arithmetic has no such limit), repeat the call on the fp32 matrix cores.  On by default for
        `from_checkpoint` (released weights were never run through this path); off for in-memory states."""
        self.check_overflow = bool(check_overflow)
        self.overflow_fallbacks = 0
        self.module_dtype = torch.float32      # torch.bfloat16 after .to(dtype=torch.bfloat16): the text2semantic CLI's mode
        self._raw_state = None                 # the state dict last loaded (kept by reference for .to(dtype=...))
        self.lib = _lib.load()
        self.config = config if isinstance(config, DacConfig) else (DacConfig.from_any(config) if config else DacConfig())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.FishmiError("MiDAC needs a GPU device (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self._c = self.config.to_c()
        need = self.lib.fmi_dac_arena_bytes(C.byref(self._c))
        if need < 0:
            check(-1)
        self.arena = torch.empty(need, dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        check(self.lib.fmi_dac_create(C.byref(self._c), C.c_void_p(self.arena.data_ptr()), need, C.byref(h)))
        self._h = h
        self.sample_rate = self.config.sample_rate
        self.frame_length = self.config.frame_length
        self.hop_length = int(math.prod(self.config.encoder_rates))
        self._dtype_probe = torch.empty(0, dtype=torch.float32, device=self.device)
        self._planes = 2                       # decode-side arithmetic outside autocast (set_precision)
        self._lock = threading.RLock()         # (precision, call) pairs of concurrent request threads stay together
        self._async = False
        self._keep_async: list = []
        self._fp32_streams: set = set()        # open streams that fell back to the fp32 matrix cores (_decode_call)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.fmi_dac_destroy(h)
            self._h = None

    def parameters(self) -> Iterable[torch.Tensor]:
        yield self._dtype_probe             # carries the module dtype, like next(codec.parameters()).dtype upstream

    def to(self, *args, **kwargs):
        """nn.Module.to for the one thing callers use it for (text2semantic/inference.py:416
        `codec.to(device=device, dtype=precision)`): dtype=torch.bfloat16 switches the whole codec to bf16-module
        numerics -- parameters rounded to bf16 (weight-norm products evaluated in bf16), every conv / linear with bf16
        operands and a bf16 result, bf16 audio in and out; torch.float32 switches back.  Elementwise ops between the
        contractions keep fp32 intermediates (torch rounds each to bf16): the same network within bf16 rounding noise,
        calibrated against the oracle in that mode in tests/test_dac_gpu.py.  fp16 is refused."""
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
        if dtype is None or dtype == self.module_dtype:
            return self
        if dtype not in (torch.float32, torch.bfloat16):
            raise _lib.FishmiError(f"MiDAC.to(dtype={dtype}): only float32 and bfloat16 are implemented")
        if self._raw_state is None:
            raise _lib.FishmiError("MiDAC.to(dtype=...): load a state dict first (the parameters are re-rounded from it)")
        self.load_folded_state(fold_weight_norm(self._raw_state, dtype))
        self.module_dtype = dtype
        self._dtype_probe = torch.empty(0, dtype=dtype, device=self.device)
        return self

    def eval(self):
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """Accepts codec.pth content (optionally wrapped / 'generator.'-prefixed like dac/inference.py:29-42)."""
        if "state_dict" in state:
            state = state["state_dict"]
        if any("generator" in k for k in state):
            state = {k.replace("generator.", ""): v for k, v in state.items() if "generator." in k}
        self._raw_state = state
        return self.load_folded_state(fold_weight_norm(state, self.module_dtype), strict=strict)

    def load_folded_state(self, folded: Dict[str, torch.Tensor], strict: bool = True):
        """Plain (already weight-norm-folded) tensors by name; see expected_state_shapes()."""
        s = self._stream()
        for name, t in folded.items():
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            dims = (C.c_int64 * t.dim())(*t.shape)
            rc = self.lib.fmi_dac_load_tensor(self._h, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), dims, 1, s)
            if rc != 0 and not strict:
                continue
            check(rc)
            torch.cuda.current_stream(self.device).synchronize()
        tab = _rope_table(32768).to(self.device)
        dims = (C.c_int64 * 2)(*tab.shape)
        check(self.lib.fmi_dac_load_tensor(self._h, b"rope_table", C.c_void_p(tab.data_ptr()), 2, dims, 1, s))
        check(self.lib.fmi_dac_finalize_weights(self._h, s))
        torch.cuda.current_stream(self.device).synchronize()
        return self

    def weights_ready(self, stream=None):
        """As MiDualAR.weights_ready: the codec's stream is ordered after `stream` (default: the current one)."""
        s = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        check(self.lib.fmi_dac_weights_ready(self._h, s))

    # ---- running beside the Dual-AR frame loop (fishmi.h: fmi_dac_set_async / fmi_dac_set_stream_options)
    def set_async(self, enable: bool):
        """encode / decode calls return once enqueued on the codec's own stream and no longer make torch's current stream
        wait; order a consumer with `wait_stream()` or wait with `synchronize()` before reading the result."""
        if self._async and not enable:
            self.synchronize()     # drains the kept temporaries and a pending overflow flag of the async calls
        check(self.lib.fmi_dac_set_async(self._h, int(bool(enable))))
        self._async = bool(enable)
        return self

    def _hold(self, *tensors):
        """Internal temporaries of a call (index copies, padded audio): alive until the codec's stream is done with them.
        Synchronous mode: the caller's stream waits for the call, so the last call's are enough.  Async mode: nothing
        orders torch's allocator after the codec stream, so every call's temporaries are kept until `synchronize()` /
        `wait_stream()`."""
        if self._async:
            self._keep_async.extend(tensors)
        else:
            self._keep = tensors

    def set_background(self, lds_floor_bytes: int):
        """Floor under the dynamic LDS of the decode-side conv kernels (> 80 KiB: one work-group per CU, so that another
        queue's work-groups can be co-resident); 0 = off."""
        check(self.lib.fmi_dac_set_background(self._h, int(lds_floor_bytes)))
        return self

    def wait_stream(self):
        """order torch's current stream after the codec calls enqueued so far (no host wait)"""
        check(self.lib.fmi_dac_wait(self._h, self._stream()))
        for t in self._keep_async:      # freed blocks are reused on torch's stream, which is now behind the codec's
            t.record_stream(torch.cuda.current_stream(self.device))
        self._keep_async = []

    def synchronize(self):
        """host wait for everything enqueued on the codec's stream.  In async mode with check_overflow this is also where
        an fp16-range overflow of the calls since the last synchronize is reported: an async call cannot be redone on
        the fp32 matrix cores behind the caller's back (its output was already handed out), so it raises."""
        check(self.lib.fmi_dac_synchronize(self._h))
        self._keep_async = []
        if self._async and self.check_overflow and self._planes == 2 and self.fp16_overflowed():
            raise _lib.FishmiError("an operand of the fp16-split codec arithmetic left the fp16 range during an async "
                                   "call; its audio is saturated -- redo the call with set_async(False) (which falls "
                                   "back to the fp32 matrix cores) or set_precision(0)")

    def set_stream_options(self, priority: int = 0, cu_mask: Optional[Sequence[int]] = None):
        """priority: -1 highest, 0 default, 1 lowest; cu_mask: 32-bit words, bit i = compute unit i may run this codec's
        kernels (overrides the priority)."""
        if cu_mask:
            words = (C.c_uint32 * len(cu_mask))(*[int(w) & 0xFFFFFFFF for w in cu_mask])
            check(self.lib.fmi_dac_set_stream_options(self._h, int(priority), len(cu_mask), words))
        else:
            check(self.lib.fmi_dac_set_stream_options(self._h, int(priority), 0, None))
        return self

    def set_precision(self, planes: int):
        """Decode-side contraction arithmetic: 2 = fp16 matrix cores on the scaled two-term operand split (fp32-class,
        the default), 0 = fp32 matrix cores (round-1 path), 1 = bf16 operands and results (what autocast selects);
        see include/fishmi.h."""
        with self._lock:
            check(self.lib.fmi_dac_set_precision(self._h, int(planes)))
            self._planes = int(planes)
        return self

    def _decode_call(self, fn, *args, stream_id: Optional[int] = None):
        """Run one decode-side entry point with the arithmetic the caller's context asks for: inside
        ``torch.autocast(device_type="cuda", dtype=torch.bfloat16)`` -- how the engine calls from_indices
        (fish_speech/inference_engine/__init__.py:179-192) -- every conv / linear rounds its operands and its result to
        bf16 with fp32 accumulation, as autocast does to F.conv1d / F.conv_transpose1d / F.linear; elementwise ops,
        norms and residual adds stay fp32 (torch's type promotion gives fp32 there too, parameters being fp32).
        Outside autocast the configured precision applies (default: the fp32-class fp16-split arithmetic).
        `stream_id`: a stream that once needed the fp32 fallback (check_overflow) stays on the fp32 matrix cores until it is
        closed -- its kept quantizer-side state was rebuilt in that arithmetic, and chunks that alternate between the two
        would neither be bit-identical to an offline decode nor cheap (every overflowing chunk recomputes the stream)."""
        planes = self._planes
        if planes == 2 and stream_id is not None and stream_id in self._fp32_streams:
            planes = 0
        if self.module_dtype == torch.bfloat16:
            planes = 1
        elif torch.is_autocast_enabled("cuda"):
            if torch.get_autocast_dtype("cuda") != torch.bfloat16:
                raise _lib.FishmiError("MiDAC supports autocast(dtype=torch.bfloat16) only (the engine's --half fp16 mode "
                                       "is not implemented)")
            planes = 1
        with self._lock:
            check(self.lib.fmi_dac_set_precision(self._h, planes))
            try:
                check(fn(self._h, *args))
                if planes == 2 and self.check_overflow and not self._async and self.fp16_overflowed():
                    # saturated samples, not NaNs -- but not the reference's either: redo on the fp32 matrix cores
                    self.overflow_fallbacks += 1
                    check(self.lib.fmi_dac_set_precision(self._h, 0))
                    planes = 0
                    if stream_id is not None:
                        self._fp32_streams.add(stream_id)
                    check(fn(self._h, *args))
            finally:
                if planes != self._planes:
                    check(self.lib.fmi_dac_set_precision(self._h, self._planes))

    def fp16_overflowed(self) -> bool:
        """Waits for the codec's stream; True if an operand of the fp16-split arithmetic left the fp16 range since the
        last call of this method (the flag is cleared)."""
        f = C.c_int(0)
        check(self.lib.fmi_dac_fp16_overflow(self._h, C.byref(f)))
        return bool(f.value)

    @classmethod
    def from_state_dict(cls, config, state, device="cuda:0") -> "MiDAC":
        return cls(config, device=device).load_state_dict(state)

    @classmethod
    def from_checkpoint(cls, path, device="cuda:0", **config_overrides) -> "MiDAC":
        """codec.pth loader of the CLIs (dac/inference.py:29-42: optional `state_dict` wrapper, `generator.` prefix
        stripped); the architecture is read off the tensor shapes."""
        state = torch.load(str(path), map_location="cpu", mmap=True, weights_only=True)
        if "state_dict" in state:
            state = state["state_dict"]
        if any("generator" in k for k in state):
            state = {k.replace("generator.", ""): v for k, v in state.items() if "generator." in k}
        folded = fold_weight_norm(state)
        m = cls(DacConfig.from_state_dict(folded, **config_overrides), device=device, check_overflow=True)
        m._raw_state = state
        return m.load_folded_state(folded)

    # ---- DAC.encode (modded_dac.py:874-923)
    @torch.no_grad()
    def encode(self, audio_data: torch.Tensor, audio_lengths: Optional[torch.Tensor] = None, n_quantizers=None, **kw):
        if audio_data.ndim == 2:
            audio_data = audio_data.unsqueeze(1)
        length = audio_data.shape[-1]
        fl = self.frame_length
        right = math.ceil(length / fl) * fl - length
        audio_data = audio_data.to(device=self.device)
        if self.module_dtype == torch.bfloat16:   # the CLI hands the bf16 module bf16 samples (inference.py:431)
            audio_data = audio_data.to(torch.bfloat16)
        audio = torch.nn.functional.pad(audio_data.to(torch.float32), (0, right)).contiguous()
        if audio_lengths is None:
            audio_lengths = torch.tensor([length + right], device=self.device, dtype=torch.long)
        B, _, N = audio.shape
        T = N // fl
        idx = torch.empty(B, self.config.n_codebooks + 1, T, dtype=torch.int64, device=self.device)
        check(self.lib.fmi_dac_encode(self._h, C.c_void_p(audio.data_ptr()), B, N, C.c_void_p(idx.data_ptr()), self._stream()))
        if self._async:
            self.wait_stream()     # idx is consumed on torch's stream (and `lens` below is computed there)
        lens = torch.ceil(audio_lengths.to(self.device) / fl).long()
        self._hold(audio)
        return idx, lens

    # ---- DAC.from_indices (modded_dac.py:925-927).  Clamps `indices` in place like rvq.py:354-359.
    @torch.no_grad()
    def from_indices(self, indices: torch.Tensor) -> torch.Tensor:
        if indices.device != self.device or indices.dtype != torch.int64 or not indices.is_contiguous():
            work = indices.to(device=self.device, dtype=torch.int64).contiguous()
        else:
            work = indices
        B, nb, T = work.shape
        if nb != self.config.n_codebooks + 1:
            raise ValueError(f"expected {self.config.n_codebooks + 1} codebooks, got {nb}")
        out = torch.empty(B, 1, T * self.frame_length, dtype=torch.float32, device=self.device)
        self._decode_call(self.lib.fmi_dac_decode, C.c_void_p(work.data_ptr()), B, T, C.c_void_p(out.data_ptr()), self._stream())
        if self._async and (work is not indices or self.module_dtype != torch.float32):
            self.wait_stream()     # the copy-back / dtype conversion below run on torch's stream: order it after the decode
        if work is not indices:
            try:
                indices.copy_(work)  # the reference mutates its argument; keep that visible to the caller
            except Exception:
                pass
        self._hold(work)
        return out.to(self.module_dtype) if self.module_dtype != torch.float32 else out

    # ---- ragged batch decode (the server's batched decode, tools/server/model_utils.py:61-86, over utterances of
    # DIFFERENT lengths): utterances are grouped `max_group` at a time in order of length, a group is padded at its END
    # to the longest member and decoded by ONE from_indices call; every codec layer is causal (modded_dac.py:521-588,
    # window mask 380-398) and a row's result does not depend on its batch, so each utterance's samples are bit-identical
    # to decoding it alone.  `pad_waste` bounds the padded frames a group may carry (fraction of its real frames); a
    # group is closed early when the next utterance would exceed it.
    @torch.no_grad()
    def from_indices_ragged(self, codes: Sequence[torch.Tensor], max_group: int = 8, pad_waste: float = 0.35) -> List[torch.Tensor]:
        """codes[i]: (1+n_codebooks, T_i) or (1, 1+n_codebooks, T_i) integer.  Returns [(1, 1, T_i * frame_length)]."""
        items = []
        for i, c in enumerate(codes):
            c = c[0] if c.ndim == 3 else c
            if c.shape[0] != self.config.n_codebooks + 1:
                raise ValueError(f"expected {self.config.n_codebooks + 1} codebooks, got {c.shape[0]}")
            items.append((int(c.shape[1]), i, c))
        out: List[Optional[torch.Tensor]] = [None] * len(items)
        order = sorted((it for it in items if it[0] > 0), key=lambda it: -it[0])
        fl = self.frame_length
        g = 0
        while g < len(order):
            tmax, real, e = order[g][0], 0, g
            while e < len(order) and e - g < max_group:
                r = real + order[e][0]
                if e > g and (e - g + 1) * tmax - r > pad_waste * r:
                    break
                real, e = r, e + 1
            group = order[g:e]
            batch = torch.zeros(len(group), self.config.n_codebooks + 1, tmax, dtype=torch.int64, device=self.device)
            for j, (t, _, c) in enumerate(group):
                batch[j, :, :t] = c.to(device=self.device, dtype=torch.int64)
            wav = self.from_indices(batch)
            for j, (t, i, _) in enumerate(group):
                out[i] = wav[j:j + 1, :, : t * fl]
            g = e
        for t, i, _ in items:
            if t == 0:
                out[i] = torch.zeros(1, 1, 0, dtype=self.module_dtype, device=self.device)
        return out  # type: ignore[return-value]

    # ---- incremental decode for streaming: audio of frames [t0, T) given all codes so far.  Bit-identical to
    # from_indices(final codes)[..., t0*frame_length : T*frame_length] because every codec layer is causal
    # (modded_dac.py:521-588; window mask 380-398).  `indices` is clamped in place like from_indices.
    # stream_id: keep the quantizer side (per-layer K/V of the windowed transformer, its output, the upsampled
    # latents) of frames [0, t0) on the device between the calls of ONE stream (fmi_dac_decode_tail_cached): the
    # caller promises that the codes of frames [0, t0) are those of its previous call with this id.
    @torch.no_grad()
    def from_indices_tail(self, indices: torch.Tensor, t0: int, stream_id: Optional[int] = None) -> torch.Tensor:
        work = indices.to(device=self.device, dtype=torch.int64).contiguous()
        B, nb, T = work.shape
        if nb != self.config.n_codebooks + 1:
            raise ValueError(f"expected {self.config.n_codebooks + 1} codebooks, got {nb}")
        if not 0 <= t0 < T:
            raise ValueError(f"t0={t0} outside [0, {T})")
        out = torch.empty(B, 1, (T - t0) * self.frame_length, dtype=torch.float32, device=self.device)
        if stream_id is None:
            self._decode_call(self.lib.fmi_dac_decode_tail, C.c_void_p(work.data_ptr()), B, T, int(t0),
                              C.c_void_p(out.data_ptr()), self._stream())
        else:
            self._decode_call(self.lib.fmi_dac_decode_tail_cached, C.c_void_p(work.data_ptr()), B, T, int(t0),
                              C.c_int64(int(stream_id)), C.c_void_p(out.data_ptr()), self._stream(),
                              stream_id=int(stream_id))
        self._hold(work)
        return out

    _stream_ids = itertools.count(1)

    @classmethod
    def new_stream_id(cls) -> int:
        return next(cls._stream_ids)

    def stream_reset(self) -> None:
        """free the device state kept for from_indices_tail(stream_id=...)"""
        check(self.lib.fmi_dac_stream_reset(self._h))
        self._fp32_streams.clear()

    def close_stream(self, stream_id: int) -> None:
        """The stream `stream_id` has ended: drop its incremental state now (fmi_dac_stream_close) instead of letting it
        age out of the 16 kept states."""
        check(self.lib.fmi_dac_stream_close(self._h, int(stream_id)))
        self._fp32_streams.discard(int(stream_id))

    @property
    def context_frames(self) -> int:
        """left context (frames) the decoder conv stack re-reads per incremental call"""
        return int(self.lib.fmi_dac_context_frames(self._h))

    # ---- DAC.decode (modded_dac.py:929-946): latent z (B, latent_dim, L) -> waveform
    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cc, L = z.shape
        if Cc != self.config.latent_dim:
            raise ValueError(f"expected {self.config.latent_dim} latent channels, got {Cc}")
        out = torch.empty(B, 1, L * self.hop_length, dtype=torch.float32, device=self.device)
        self._decode_call(self.lib.fmi_dac_decode_latent, C.c_void_p(z.data_ptr()), B, L, C.c_void_p(out.data_ptr()), self._stream())
        self._hold(z)
        return out

    def debug_z(self, B: int) -> torch.Tensor:
        """quantizer.decode output (B, latent_dim, 4T) of the last from_indices call (parity tap)."""
        from .dual_ar import _from_ptr

        p, cc, ll = C.c_void_p(), C.c_int(), C.c_int()
        check(self.lib.fmi_dac_debug_z(self._h, C.byref(p), C.byref(cc), C.byref(ll)))
        torch.cuda.synchronize(self.device)
        return _from_ptr(p.value, (B, cc.value, ll.value), torch.float32, self.device).clone()

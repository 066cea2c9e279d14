"""CPU oracle for the modded-DAC codec (encode / from_indices).  TEST INFRASTRUCTURE ONLY.

Torch-CPU restatement of ``fish_speech/models/dac/modded_dac.py`` and ``rvq.py`` plus the few
functions they import from ``descript-audio-codec`` 1.0.0 (Snake1d, weight-normed convs,
VectorQuantize, ResidualVectorQuantize -- restated from the published source; the wheel is not in
this image, so THAT part of the parity is unpinned, see oracle/README.md).

Pinned against the unmodified reference ``DAC`` (imported with oracle/stubs) by
``tests/test_oracle_cpu.py`` where the reference checkout exists, and against fixtures it produced
(tests/golden/dac_*.npz, oracle/gen_golden_dac.py).

State dict = the keys of ``codec.pth``: weight-normed convs carry
``parametrizations.weight.original0/1`` (new API, modded_dac.py:554-556) or ``weight_g/weight_v``
(old API, third-party quantizer in/out_proj); ``fold_weight_norm`` turns both into plain weights
with the same ``torch._weight_norm`` the reference's parametrization calls.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class DacConfig:
    """fish_speech/configs/modded_dac_vq.yaml; small values give test-sized models."""

    encoder_dim: int = 64
    encoder_rates: Tuple[int, ...] = (2, 4, 8, 8)
    decoder_dim: int = 1536
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    n_codebooks: int = 9
    codebook_size: int = 1024
    semantic_codebook_size: int = 4096
    codebook_dim: int = 8
    downsample: Tuple[int, ...] = (2, 2)
    tf_layers: int = 8            # quantizer pre_module / post_module (yaml 30-48)
    tf_ffn_mult: int = 3          # intermediate_size 3072 = 3 * 1024
    tf_window: int = 128
    enc_tf_layers: int = 4        # encoder_transformer_layers [0,0,0,4]
    enc_tf_window: int = 512      # modded_dac.py:641 default
    head_dim: int = 64
    rope_base: float = 10000.0
    norm_eps: float = 1e-5
    sample_rate: int = 44100

    @property
    def latent_dim(self) -> int:
        return self.encoder_dim * (2 ** len(self.encoder_rates))  # modded_dac.py:828-829

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.encoder_rates))

    @property
    def frame_length(self) -> int:
        return self.hop_length * int(math.prod(self.downsample))  # modded_dac.py:861


def small_config() -> DacConfig:
    """Test-sized codec: same topology, ~1.5 M parameters."""
    return DacConfig(encoder_dim=8, decoder_dim=96, n_codebooks=3, codebook_size=64, semantic_codebook_size=128,
                     tf_layers=2, tf_window=8, enc_tf_layers=2, enc_tf_window=16)


# ----------------------------------------------------------------------------- state dict layout


def _tf_shapes(prefix: str, dim: int, n_layer: int, ffn: int, out: Dict[str, tuple]):
    for i in range(n_layer):
        p = f"{prefix}.layers.{i}"
        out[f"{p}.attention.wqkv.weight"] = (3 * dim, dim)
        out[f"{p}.attention.wo.weight"] = (dim, dim)
        out[f"{p}.feed_forward.w1.weight"] = (ffn, dim)
        out[f"{p}.feed_forward.w3.weight"] = (ffn, dim)
        out[f"{p}.feed_forward.w2.weight"] = (dim, ffn)
        out[f"{p}.ffn_norm.weight"] = (dim,)
        out[f"{p}.attention_norm.weight"] = (dim,)
        out[f"{p}.attention_layer_scale.gamma"] = (dim,)
        out[f"{p}.ffn_layer_scale.gamma"] = (dim,)
    out[f"{prefix}.norm.weight"] = (dim,)


def _wn_conv(prefix: str, cout: int, cin: int, k: int, out: Dict[str, tuple], transposed: bool = False):
    # CausalConvNet.weight_norm -> parametrizations (modded_dac.py:554-556); norm over dims != 0
    shape = (cin, cout, k) if transposed else (cout, cin, k)
    out[f"{prefix}.conv.bias"] = (cout,)
    out[f"{prefix}.conv.parametrizations.weight.original0"] = (shape[0], 1, 1)
    out[f"{prefix}.conv.parametrizations.weight.original1"] = shape


def _res_unit(prefix: str, dim: int, out: Dict[str, tuple]):
    out[f"{prefix}.block.0.alpha"] = (1, dim, 1)
    _wn_conv(f"{prefix}.block.1", dim, dim, 7, out)
    out[f"{prefix}.block.2.alpha"] = (1, dim, 1)
    _wn_conv(f"{prefix}.block.3", dim, dim, 1, out)


def state_shapes(cfg: DacConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    # encoder (modded_dac.py:670-709)
    d = cfg.encoder_dim
    _wn_conv("encoder.block.0", d, 1, 7, s)
    for bi, stride in enumerate(cfg.encoder_rates):
        d *= 2
        p = f"encoder.block.{bi + 1}"
        for r in range(3):
            _res_unit(f"{p}.block.{r}", d // 2, s)
        s[f"{p}.block.3.alpha"] = (1, d // 2, 1)
        _wn_conv(f"{p}.block.4", d, d // 2, 2 * stride, s)
        if bi == len(cfg.encoder_rates) - 1 and cfg.enc_tf_layers:
            _tf_shapes(f"{p}.block.5", d, cfg.enc_tf_layers, d * 3, s)
    nb = len(cfg.encoder_rates) + 1
    s[f"encoder.block.{nb}.alpha"] = (1, d, 1)
    _wn_conv(f"encoder.block.{nb + 1}", cfg.latent_dim, d, 3, s)
    # quantizer (rvq.py:204-291)
    L = cfg.latent_dim
    for name, n, size in (("semantic_quantizer", 1, cfg.semantic_codebook_size), ("quantizer", cfg.n_codebooks, cfg.codebook_size)):
        for i in range(n):
            p = f"quantizer.{name}.quantizers.{i}"
            s[f"{p}.in_proj.bias"] = (cfg.codebook_dim,)
            s[f"{p}.in_proj.weight_g"] = (cfg.codebook_dim, 1, 1)
            s[f"{p}.in_proj.weight_v"] = (cfg.codebook_dim, L, 1)
            s[f"{p}.out_proj.bias"] = (L,)
            s[f"{p}.out_proj.weight_g"] = (L, 1, 1)
            s[f"{p}.out_proj.weight_v"] = (L, cfg.codebook_dim, 1)
            s[f"{p}.codebook.weight"] = (size, cfg.codebook_dim)
    for name in ("downsample", "upsample"):
        for i, f in enumerate(cfg.downsample):
            p = f"quantizer.{name}.{i}"
            fac = f if name == "downsample" else list(reversed(cfg.downsample))[i]
            s[f"{p}.0.conv.weight"] = (L, L, fac)  # plain conv / conv-transpose, no weight norm
            s[f"{p}.0.conv.bias"] = (L,)
            s[f"{p}.1.gamma"] = (L,)
            s[f"{p}.1.dwconv.conv.weight"] = (L, 1, 7)
            s[f"{p}.1.dwconv.conv.bias"] = (L,)
            s[f"{p}.1.norm.weight"] = (L,)
            s[f"{p}.1.norm.bias"] = (L,)
            s[f"{p}.1.pwconv1.weight"] = (4 * L, L)
            s[f"{p}.1.pwconv1.bias"] = (4 * L,)
            s[f"{p}.1.pwconv2.weight"] = (L, 4 * L)
            s[f"{p}.1.pwconv2.bias"] = (L,)
    for name in ("pre_module", "post_module"):
        _tf_shapes(f"quantizer.{name}", L, cfg.tf_layers, L * cfg.tf_ffn_mult, s)
    # decoder (modded_dac.py:760-801)
    _wn_conv("decoder.model.0", cfg.decoder_dim, L, 7, s)
    for i, stride in enumerate(cfg.decoder_rates):
        cin, cout = cfg.decoder_dim // 2 ** i, cfg.decoder_dim // 2 ** (i + 1)
        p = f"decoder.model.{i + 1}"
        s[f"{p}.block.0.alpha"] = (1, cin, 1)
        _wn_conv(f"{p}.block.1", cout, cin, 2 * stride, s, transposed=True)
        for r in range(3):
            _res_unit(f"{p}.block.{2 + r}", cout, s)
    nd = len(cfg.decoder_rates) + 1
    cl = cfg.decoder_dim // 2 ** len(cfg.decoder_rates)
    s[f"decoder.model.{nd}.alpha"] = (1, cl, 1)
    _wn_conv(f"decoder.model.{nd + 1}", 1, cl, 7, s)
    return s


def make_synthetic_state(cfg: DacConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights.  Init scales follow the reference (trunc_normal 0.02, modded_dac.py:455-458)
    widened so that every branch matters: Snake alpha ~ U(0.5,1.5), LayerScale / ConvNeXt gamma ~ 0.3
    (their defaults 1e-2 / 1e-6 would hide the branches), weight-norm g ~ ||v||."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in state_shapes(cfg).items():
        if name.endswith("alpha"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith(".gamma"):
            t = 0.3 * (1 + 0.2 * torch.randn(shape, generator=g))
        elif name.endswith("norm.weight") or name.endswith("_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("norm.bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("original0") or name.endswith("weight_g"):
            t = None  # filled below from v
        elif name.endswith("codebook.weight"):
            t = torch.randn(shape, generator=g)
        else:
            fan_in = int(math.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
        out[name] = t
    for name in list(out):
        if out[name] is None:
            vname = name.replace("original0", "original1").replace("weight_g", "weight_v")
            v = out[vname]
            nrm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
            # gain 0.75: with 1.0 the 12 residual units of the decoder amplify ~2.5x per stage and the tanh
            # saturates; 0.75 keeps the waveform inside (-1, 1) with RMS ~0.2 (audio-like) at both sizes
            out[name] = 0.75 * nrm * (1 + 0.1 * torch.randn(nrm.shape, generator=g))
    return out


def fold_weight_norm(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """w = g * v / ||v|| (norm over all dims but 0) via torch._weight_norm, the function both
    weight-norm APIs of the reference end up calling."""
    out = {}
    for k, v in state.items():
        if k.endswith("parametrizations.weight.original1"):
            base = k[: -len("parametrizations.weight.original1")]
            out[base + "weight"] = torch._weight_norm(v, state[base + "parametrizations.weight.original0"], 0)
        elif k.endswith("weight_v"):
            base = k[: -len("weight_v")]
            out[base + "weight"] = torch._weight_norm(v, state[base + "weight_g"], 0)
        elif k.endswith("parametrizations.weight.original0") or k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out


# ----------------------------------------------------------------------------- primitives


def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """dac.nn.layers.snake: x + (alpha + 1e-9)^-1 * sin(alpha x)^2, alpha (1,C,1)."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def causal_conv(x, w, b, stride=1, dilation=1, groups=1):
    """CausalConvNet.forward (modded_dac.py:521-552): left pad k_eff - stride, right pad so that the
    last window is complete, then a plain conv1d."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    pad = k_eff - stride
    length = x.shape[-1]
    n_frames = (length - k_eff + pad) / stride + 1
    extra = (math.ceil(n_frames) - 1) * stride + (k_eff - pad) - length
    x = F.pad(x, (pad, extra))
    return F.conv1d(x, w, b, stride=stride, dilation=dilation, groups=groups)


def causal_conv_transpose(x, w, b, stride):
    """CausalTransConvNet.forward (modded_dac.py:563-582): conv_transpose1d, drop k - stride samples
    on the right."""
    y = F.conv_transpose1d(x, w, b, stride=stride)
    pad = w.shape[-1] - stride
    return y[..., : y.shape[-1] - pad] if pad > 0 else y


def rope_table(seq_len: int, n_elem: int, base: float) -> torch.Tensor:
    """modded_dac.py:442-452: bf16 by default."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(seq_len), freqs)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(torch.bfloat16)


def apply_rope(x, tab):
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    t = tab.view(1, xs.size(1), 1, xs.size(3), 2)
    re = xs[..., 0] * t[..., 0] - xs[..., 1] * t[..., 1]
    im = xs[..., 1] * t[..., 0] + xs[..., 0] * t[..., 1]
    return torch.stack([re, im], dim=-1).flatten(3).type_as(x)


def rms_norm(x, w, eps):
    xf = x.float()
    return (xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)).type_as(x) * w


class DacOracle:
    def __init__(self, cfg: DacConfig, state: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.w = fold_weight_norm(state)
        self._rope: Dict[int, torch.Tensor] = {}

    # ---- transformer (modded_dac.py:97-439)
    def window_transformer(self, x: torch.Tensor, prefix: str, n_layer: int, window: int) -> torch.Tensor:
        """WindowLimitedTransformer.forward, channels-first in/out.  Query i sees keys
        max(0, i-window+1) .. i (modded_dac.py:380-398)."""
        w, cfg = self.w, self.cfg
        x = x.transpose(1, 2)
        B, T, dim = x.shape
        hd = cfg.head_dim
        nh = dim // hd
        if T not in self._rope:
            self._rope[T] = rope_table(T, hd, cfg.rope_base)
        tab = self._rope[T]
        i = torch.arange(T).view(-1, 1)
        j = torch.arange(T).view(1, -1)
        mask = ((j <= i) & (j >= (i - window + 1).clamp(min=0)))[None, None]
        for li in range(n_layer):
            p = f"{prefix}.layers.{li}"
            h_in = rms_norm(x, w[f"{p}.attention_norm.weight"], cfg.norm_eps)
            q, k, v = F.linear(h_in, w[f"{p}.attention.wqkv.weight"]).split([dim, dim, dim], dim=-1)
            q = apply_rope(q.view(B, T, nh, hd), tab).transpose(1, 2)
            k = apply_rope(k.view(B, T, nh, hd), tab).transpose(1, 2)
            v = v.view(B, T, nh, hd).transpose(1, 2)
            y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            y = y.transpose(1, 2).contiguous().view(B, T, dim)
            h = x + F.linear(y, w[f"{p}.attention.wo.weight"]) * w[f"{p}.attention_layer_scale.gamma"]
            f_in = rms_norm(h, w[f"{p}.ffn_norm.weight"], cfg.norm_eps)
            ff = F.linear(F.silu(F.linear(f_in, w[f"{p}.feed_forward.w1.weight"]))
                          * F.linear(f_in, w[f"{p}.feed_forward.w3.weight"]), w[f"{p}.feed_forward.w2.weight"])
            x = h + ff * w[f"{p}.ffn_layer_scale.gamma"]
        x = rms_norm(x, w[f"{prefix}.norm.weight"], cfg.norm_eps)
        return x.transpose(1, 2)

    # ---- conv blocks
    def _conv(self, x, prefix, **kw):
        return causal_conv(x, self.w[f"{prefix}.conv.weight"], self.w[f"{prefix}.conv.bias"], **kw)

    def res_unit(self, x, prefix, dilation):
        """ResidualUnit (modded_dac.py:599-620): x + conv1(snake(conv7_dil(snake(x))))."""
        w = self.w
        y = snake(x, w[f"{prefix}.block.0.alpha"])
        y = self._conv(y, f"{prefix}.block.1", dilation=dilation)
        y = snake(y, w[f"{prefix}.block.2.alpha"])
        y = self._conv(y, f"{prefix}.block.3")
        return x + y

    def convnext(self, x, prefix):
        """ConvNeXtBlock (rvq.py:129-191)."""
        w = self.w
        L = x.shape[1]
        y = causal_conv(x, w[f"{prefix}.dwconv.conv.weight"], w[f"{prefix}.dwconv.conv.bias"], groups=L)
        y = y.permute(0, 2, 1)
        y = F.layer_norm(y, (L,), w[f"{prefix}.norm.weight"], w[f"{prefix}.norm.bias"], 1e-6)
        y = F.linear(y, w[f"{prefix}.pwconv1.weight"], w[f"{prefix}.pwconv1.bias"])
        y = F.gelu(y)
        y = F.linear(y, w[f"{prefix}.pwconv2.weight"], w[f"{prefix}.pwconv2.bias"])
        y = w[f"{prefix}.gamma"] * y
        return x + y.permute(0, 2, 1)

    # ---- encoder / decoder (modded_dac.py:670-801)
    def encoder(self, x):
        cfg, w = self.cfg, self.w
        x = self._conv(x, "encoder.block.0")
        for bi, stride in enumerate(cfg.encoder_rates):
            p = f"encoder.block.{bi + 1}"
            for r, dil in enumerate((1, 3, 9)):
                x = self.res_unit(x, f"{p}.block.{r}", dil)
            x = snake(x, w[f"{p}.block.3.alpha"])
            x = self._conv(x, f"{p}.block.4", stride=stride)
            if bi == len(cfg.encoder_rates) - 1 and cfg.enc_tf_layers:
                x = self.window_transformer(x, f"{p}.block.5", cfg.enc_tf_layers, cfg.enc_tf_window)
        nb = len(cfg.encoder_rates) + 1
        x = snake(x, w[f"encoder.block.{nb}.alpha"])
        return self._conv(x, f"encoder.block.{nb + 1}")

    def decoder(self, z):
        cfg, w = self.cfg, self.w
        x = self._conv(z, "decoder.model.0")
        for i, stride in enumerate(cfg.decoder_rates):
            p = f"decoder.model.{i + 1}"
            x = snake(x, w[f"{p}.block.0.alpha"])
            x = causal_conv_transpose(x, w[f"{p}.block.1.conv.weight"], w[f"{p}.block.1.conv.bias"], stride)
            for r, dil in enumerate((1, 3, 9)):
                x = self.res_unit(x, f"{p}.block.{2 + r}", dil)
        nd = len(cfg.decoder_rates) + 1
        x = snake(x, w[f"decoder.model.{nd}.alpha"])
        return torch.tanh(self._conv(x, f"decoder.model.{nd + 1}"))

    # ---- quantizer (rvq.py:293-366 + dac.nn.quantize)
    def _vq(self, prefix: str, residual: torch.Tensor):
        """VectorQuantize.forward at inference: in_proj -> cosine-nearest code -> straight-through
        sum -> out_proj.  Returns (z_q_i, indices)."""
        w = self.w
        z_e = F.conv1d(residual, w[f"{prefix}.in_proj.weight"], w[f"{prefix}.in_proj.bias"])
        B, Dc, T = z_e.shape
        enc = F.normalize(z_e.permute(0, 2, 1).reshape(B * T, Dc))
        cb = F.normalize(w[f"{prefix}.codebook.weight"])
        dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
        idx = (-dist).max(1)[1].reshape(B, T)
        z_q = F.embedding(idx, w[f"{prefix}.codebook.weight"]).transpose(1, 2)
        z_q = z_e + (z_q - z_e)  # straight-through estimator, kept by the reference at inference
        return F.conv1d(z_q, w[f"{prefix}.out_proj.weight"], w[f"{prefix}.out_proj.bias"]), idx

    def quantize(self, z: torch.Tensor):
        """DownsampleResidualVectorQuantize.forward up to the codes (rvq.py:293-316); the reference
        also runs post_module + upsample here and discards the result (modded_dac.py:921)."""
        cfg, w = self.cfg, self.w
        for i, fac in enumerate(cfg.downsample):
            p = f"quantizer.downsample.{i}"
            z = causal_conv(z, w[f"{p}.0.conv.weight"], w[f"{p}.0.conv.bias"], stride=fac)
            z = self.convnext(z, f"{p}.1")
        z = self.window_transformer(z, "quantizer.pre_module", cfg.tf_layers, cfg.tf_window)
        sem_z, sem_idx = self._vq("quantizer.semantic_quantizer.quantizers.0", z)
        residual = z - sem_z
        codes = [sem_idx]
        for i in range(cfg.n_codebooks):
            zq_i, idx = self._vq(f"quantizer.quantizer.quantizers.{i}", residual)
            residual = residual - zq_i
            codes.append(idx)
        return torch.stack(codes, dim=1), z

    def dequantize(self, indices: torch.Tensor) -> torch.Tensor:
        """DownsampleResidualVectorQuantize.decode (rvq.py:352-366).  Clamps IN PLACE like the reference."""
        cfg, w = self.cfg, self.w
        indices[:, 0] = torch.clamp(indices[:, 0], max=cfg.semantic_codebook_size - 1)
        indices[:, 1:] = torch.clamp(indices[:, 1:], max=cfg.codebook_size - 1)

        def from_codes(name, codes):
            z_q = 0.0
            for i in range(codes.shape[1]):
                p = f"quantizer.{name}.quantizers.{i}"
                z_p = F.embedding(codes[:, i], w[f"{p}.codebook.weight"]).transpose(1, 2)
                z_q = z_q + F.conv1d(z_p, w[f"{p}.out_proj.weight"], w[f"{p}.out_proj.bias"])
            return z_q

        z = from_codes("semantic_quantizer", indices[:, :1]) + from_codes("quantizer", indices[:, 1:])
        z = self.window_transformer(z, "quantizer.post_module", cfg.tf_layers, cfg.tf_window)
        ups = list(reversed(list(enumerate(cfg.downsample))))
        for i, (_, fac) in enumerate(ups):
            p = f"quantizer.upsample.{i}"
            z = causal_conv_transpose(z, w[f"{p}.0.conv.weight"], w[f"{p}.0.conv.bias"], fac)
            z = self.convnext(z, f"{p}.1")
        return z

    # ---- DAC.encode / from_indices (modded_dac.py:874-927)
    def encode(self, audio: torch.Tensor, audio_lengths: torch.Tensor = None):
        cfg = self.cfg
        if audio.ndim == 2:
            audio = audio.unsqueeze(1)
        length = audio.shape[-1]
        right = math.ceil(length / cfg.frame_length) * cfg.frame_length - length
        audio = F.pad(audio, (0, right))
        if audio_lengths is None:
            audio_lengths = torch.LongTensor([length + right])
        z = self.encoder(audio)
        codes, _ = self.quantize(z)
        return codes, torch.ceil(audio_lengths / cfg.frame_length).long()

    def from_indices(self, indices: torch.Tensor) -> torch.Tensor:
        return self.decoder(self.dequantize(indices))


def make_codes(cfg: DacConfig, B: int, T: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    c = torch.empty(B, 1 + cfg.n_codebooks, T, dtype=torch.int64)
    c[:, 0] = torch.randint(0, cfg.semantic_codebook_size, (B, T), generator=g)
    c[:, 1:] = torch.randint(0, cfg.codebook_size, (B, cfg.n_codebooks, T), generator=g)
    return c

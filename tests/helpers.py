"""Shared helpers for the tests: golden-case loading and tolerant bf16 comparison."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import dual_ar as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_dualar_case(name: str):
    z = np.load(os.path.join(GOLDEN, f"dualar_{name}.npz"))
    kw = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        kw[str(k)] = int(v)
    cfg = O.DualARConfig(**kw)
    state = O.make_synthetic_state(cfg, seed=int(z["state_seed"]), head_gain=float(z["head_gain"]))
    return cfg, state, z


def bf16_close(a: torch.Tensor, b: torch.Tensor, ulps: float = 2.0, atol: float = 1e-3):
    """|a-b| <= ulps * 2^-8 * max(|a|,|b|) + atol elementwise (bf16 has 8 significant bits)."""
    a, b = a.float().cpu(), b.float().cpu()
    tol = ulps * (2.0 ** -8) * torch.maximum(a.abs(), b.abs()) + atol
    bad = (a - b).abs() > tol
    return (not bool(bad.any())), float((a - b).abs().max()), int(bad.sum())

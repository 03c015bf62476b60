"""Closed-form, RNG-free synthetic weights and inputs.

There is no network and no checkpoint (the reference's are private,
src/Experiments.ipynb:99-106), so benchmarks, parity tests and the golden
fixtures all use tensors that any machine regenerates bit-identically from
integer arithmetic: element `i` of tensor `name` is a 64-bit mix
(splitmix64 finaliser) of `(i, crc32(name), seed)` mapped to a uniform in
[-1, 1) and scaled to the variance the reference's initialisers would give.

A pure `sin(a*i+b)` recipe (SURVEY.md §8c suggestion) was rejected: it makes
every matrix rank 2, which hides layout bugs in GEMM tiles.

Key set and shapes = the reference `state_dict` minus the `nac.*` sub-tree
(reference src/model/DiTTO.py:36-64, src/components/DiT.py:78-98).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

from .config import DiTTOConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(shape, name: str, seed: int = 0) -> np.ndarray:
    """float64 uniform in [-1, 1), a pure function of (index, name, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) | (int(seed) & 0xFFFFFFFF) << 32)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(idx ^ _splitmix64(np.full(1, key, dtype=np.uint64)))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))  # [0,1)
    return (2.0 * u - 1.0).reshape(shape)


def hash_normal(shape, name: str, seed: int = 0) -> torch.Tensor:
    """fp32 ~N(0,1): Irwin-Hall sum of 4 hashed uniforms (exact integer hash + 3 adds:
    no libm call, so identical on every host)."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += hash_uniform(shape, f"{name}#{k}", seed)
    # var of U(-1,1) = 1/3 ; sum of 4 -> 4/3
    return torch.from_numpy((acc * np.sqrt(3.0 / 4.0)).astype(np.float32))


def _uniform_std(shape, name, seed, std):
    # U(-a, a) has std a/sqrt(3)
    return torch.from_numpy((hash_uniform(shape, name, seed) * (std * np.sqrt(3.0))).astype(np.float32))


def expected_state_shapes(cfg: DiTTOConfig) -> "OrderedDict[str, tuple]":
    """The DiT-stack part of the reference state_dict: key -> shape, in module order."""
    d, td, dt, S, dh = cfg.hidden_dim, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps, cfg.head_dim
    s = OrderedDict()
    s["alphas_cumprod"] = (S,)
    s["t_embedding.weight"] = (S, td)
    for i in (0, 2):
        s[f"time_embed.{i}.weight"] = (td, td)
        s[f"time_embed.{i}.bias"] = (td,)
    s["ada_ln.time_mlp.1.weight"] = (2 * d, td)
    s["ada_ln.time_mlp.1.bias"] = (2 * d,)
    s["ada_ln.text_mlp.1.weight"] = (2 * d, dt)
    s["ada_ln.text_mlp.1.bias"] = (2 * d,)
    for l in range(cfg.num_layers):
        p = f"blocks.{l}."
        s[p + "norm1.weight"] = (d,)
        s[p + "norm1.bias"] = (d,)
        s[p + "attn.in_proj_weight"] = (3 * d, d)
        s[p + "attn.in_proj_bias"] = (3 * d,)
        s[p + "attn.out_proj.weight"] = (d, d)      # dead parameter
        s[p + "attn.out_proj.bias"] = (d,)          # dead parameter
        s[p + "rotary.inv_freq"] = (dh // 2,)
        s[p + "norm2.weight"] = (d,)
        s[p + "norm2.bias"] = (d,)
        s[p + "cross_attn.in_proj_weight"] = (3 * d, d)
        s[p + "cross_attn.in_proj_bias"] = (3 * d,)
        s[p + "cross_attn.out_proj.weight"] = (d, d)
        s[p + "cross_attn.out_proj.bias"] = (d,)
        s[p + "norm3.weight"] = (d,)
        s[p + "norm3.bias"] = (d,)
        s[p + "mlp_fc1.weight"] = (4 * d, d)
        s[p + "mlp_fc1.bias"] = (4 * d,)
        s[p + "gate.weight"] = (4 * d, d)
        s[p + "gate.bias"] = (4 * d,)
        s[p + "mlp_fc2.weight"] = (d, 4 * d)
        s[p + "mlp_fc2.bias"] = (d,)
    s["proj_in.weight"] = (d, d)
    s["proj_in.bias"] = (d,)
    s["proj_out.weight"] = (d, d)
    s["proj_out.bias"] = (d,)
    s["rotary.inv_freq"] = (dh // 2,)
    return s


def cosine_betas(timesteps: int, s: float = 0.008) -> torch.Tensor:
    """Clipped betas of the cosine schedule, same torch op sequence as the reference
    (src/model/DiTTO.py:96-104) so the fp32 values are bit-identical."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0.0001, 0.9999)


def synthetic_state_dict(cfg: DiTTOConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU tensors for every key of `expected_state_shapes`."""
    sd = OrderedDict()
    dh = cfg.head_dim
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dh, 2).float() / dh))  # src/components/DiT.py:49
    for k, shp in expected_state_shapes(cfg).items():
        if k == "alphas_cumprod":
            # bug-for-bug: the buffer holds the clipped betas
            sd[k] = cosine_betas(cfg.diffusion_steps)
        elif k.endswith("inv_freq"):
            sd[k] = inv_freq.clone()
        elif k == "t_embedding.weight":
            sd[k] = _uniform_std(shp, k, seed, 1.0)            # nn.Embedding init is N(0,1)
        elif ".norm" in k and k.endswith(".weight"):
            sd[k] = 1.0 + _uniform_std(shp, k, seed, 0.1)
        elif k.endswith("bias"):
            sd[k] = _uniform_std(shp, k, seed, 0.05)
        else:  # matrices [out, in]
            fan_in = shp[-1]
            std = 1.0 / np.sqrt(fan_in)
            if "mlp_fc2" in k or "out_proj" in k or k.startswith("proj_out"):
                std *= 0.7  # keep the residual stream from blowing up over 12-24 layers
            sd[k] = _uniform_std(shp, k, seed, std)
    return sd


def synthetic_inputs(cfg: DiTTOConfig, B: int, N: int, T: int, seed: int = 7):
    """x ~N(0,1) [B,N,d], text ~N(0,1) [B,T,d_text], t int64 [B] spread over the schedule."""
    x = hash_normal((B, N, cfg.hidden_dim), "x", seed)
    text = hash_normal((B, T, cfg.text_dim), "text", seed)
    S = cfg.diffusion_steps
    t = torch.tensor([(S - 1 - (b * 7919) % S) % S for b in range(B)], dtype=torch.int64)
    return x, text, t


def slp_state_shapes(d: int, nhead: int, num_layers: int, num_classes: int) -> "OrderedDict[str, tuple]":
    """state_dict keys of the reference SLP's decoder stack + head (src/model/SpeechLP.py:22-34:
    nn.TransformerDecoderLayer(d, nhead, dim_feedforward = d * nhead), nn.Linear(d, num_classes))."""
    ff = d * nhead
    shp = OrderedDict()
    for l in range(num_layers):
        p = f"transformer.layers.{l}."
        for a in ("self_attn", "multihead_attn"):
            shp[p + a + ".in_proj_weight"] = (3 * d, d)
            shp[p + a + ".in_proj_bias"] = (3 * d,)
            shp[p + a + ".out_proj.weight"] = (d, d)
            shp[p + a + ".out_proj.bias"] = (d,)
        shp[p + "linear1.weight"] = (ff, d)
        shp[p + "linear1.bias"] = (ff,)
        shp[p + "linear2.weight"] = (d, ff)
        shp[p + "linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3"):
            shp[p + n + ".weight"] = (d,)
            shp[p + n + ".bias"] = (d,)
    shp["length_predictor.weight"] = (num_classes, d)
    shp["length_predictor.bias"] = (num_classes,)
    return shp


def synthetic_slp_state_dict(d: int, nhead: int, num_layers: int, num_classes: int, seed: int = 0):
    """Closed-form fp32 weights for the SLP decoder stack (same recipe as synthetic_state_dict)."""
    sd = OrderedDict()
    for k, shp in slp_state_shapes(d, nhead, num_layers, num_classes).items():
        if ".norm" in k and k.endswith(".weight"):
            sd[k] = 1.0 + _uniform_std(shp, "slp." + k, seed, 0.1)
        elif k.endswith("bias"):
            sd[k] = _uniform_std(shp, "slp." + k, seed, 0.05)
        else:
            sd[k] = _uniform_std(shp, "slp." + k, seed, 1.0 / np.sqrt(shp[-1]))
    return sd

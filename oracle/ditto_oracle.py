"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not product code.

CPU fp32 restatement of the reference's DiT denoise path, written as plain
functions over a `state_dict`-keyed mapping of tensors.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the reported CPU baseline.  The product
(`ditto_tts_amd/`) never imports it and has no CPU fallback.

Parity status: PINNED BY IMPORT.  The reference ships no tests or golden vectors
(SURVEY.md §4), so this restatement is pinned by executing the reference's own
modules in the build container on synthetic inputs (tests/golden/make_golden.py,
which imports /root/reference/src) and committing the outputs under
tests/golden/*.npz; tests/test_oracle_golden.py checks this file against those
fixtures everywhere, and tests/test_oracle_vs_reference.py checks it against the
live reference wherever /root/reference exists.  `SpeechGenerator` cannot be
imported (torchaudio / BigVGAN missing), so its ~20 lines of sampler arithmetic
(src/model/SpeechGenerator.py:70-72,130-164) are restated here and pinned only
through the imported `DiTTO.forward` + this restated update ("parity unpinned by
the reference" for those 4 elementwise lines).

Third-party arithmetic on the path: torch (reference pins torch==2.5.1,
requirements.txt:62; this image has 2.10.0) — F.linear, softmax, layer_norm,
gelu(erf), sigmoid, silu and nn.MultiheadAttention, whose math is restated from
torch/nn/functional.py `multi_head_attention_forward` (packed in-projection,
q * sqrt(1/E_head), bmm, softmax, bmm, out-proj).

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-5  # nn.LayerNorm default, src/components/DiT.py:23,84,89,94


# --------------------------------------------------------------------------- #
# RotaryEmbedding                                     src/components/DiT.py:43-72
# --------------------------------------------------------------------------- #
def rotary_inv_freq(head_dim: int) -> Tensor:
    """src/components/DiT.py:49"""
    return 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))


def rotary_table(inv_freq: Tensor, seq_len: int) -> Tensor:
    """RotaryEmbedding.forward, src/components/DiT.py:56-59 -> [seq_len, head_dim]."""
    t = torch.arange(seq_len).type_as(inv_freq)
    freqs = torch.outer(t, inv_freq)                       # einsum("i,j->ij")
    return torch.cat((freqs, freqs), dim=-1)


def rotate_half(x: Tensor) -> Tensor:
    """src/components/DiT.py:52-54"""
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(pos: Tensor, t: Tensor) -> Tensor:
    """src/components/DiT.py:61-72; t is [B, N, H, dh], pos is [N, dh]."""
    pos = pos.unsqueeze(0).unsqueeze(2)
    return t * pos.cos() + rotate_half(t) * pos.sin()


# --------------------------------------------------------------------------- #
# GlobalAdaLN                                          src/components/DiT.py:8-40
# --------------------------------------------------------------------------- #
def global_adaln(sd: Mapping[str, Tensor], x: Tensor, time_emb: Tensor, text_emb: Tensor,
                 prefix: str = "ada_ln.") -> Tensor:
    pooled = torch.mean(text_emb, dim=1)                                              # :27 (no mask)
    tm = F.linear(F.silu(time_emb), sd[prefix + "time_mlp.1.weight"], sd[prefix + "time_mlp.1.bias"])
    xm = F.linear(F.silu(pooled), sd[prefix + "text_mlp.1.weight"], sd[prefix + "text_mlp.1.bias"])
    time_scale, time_shift = tm.chunk(2, dim=-1)                                      # :30
    text_scale, text_shift = xm.chunk(2, dim=-1)                                      # :31
    scale = 1 + time_scale + text_scale                                               # :34
    shift = time_shift + text_shift                                                   # :35
    x = F.layer_norm(x, (x.shape[-1],), None, None, LN_EPS)                           # :38 (no affine)
    return x * scale.unsqueeze(1) + shift.unsqueeze(1)                                # :39


# --------------------------------------------------------------------------- #
# DiT block                                          src/components/DiT.py:100-157
# --------------------------------------------------------------------------- #
def dit_block(sd: Mapping[str, Tensor], prefix: str, x: Tensor, text_emb: Tensor, rotary_pos: Tensor,
              num_heads: int, taps: Optional[Dict[str, Tensor]] = None,
              dropout_p: float = 0.0, drop_mask: Optional[Tensor] = None) -> Tensor:
    B, N, d = x.shape
    dh = d // num_heads
    T = text_emb.shape[1]

    # ---- self-attention with RoPE (:103-139); NO out_proj ----
    residual = x
    u = F.layer_norm(x, (d,), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], LN_EPS)
    w, b = sd[prefix + "attn.in_proj_weight"], sd[prefix + "attn.in_proj_bias"]
    q = F.linear(u, w[:d], b[:d])                                                     # :112
    k = F.linear(u, w[d:2 * d], b[d:2 * d])                                           # :113
    v = F.linear(u, w[2 * d:], b[2 * d:])                                             # :114
    q = q.view(B, N, num_heads, dh)                                                   # :117-119
    k = k.view(B, N, num_heads, dh)
    v = v.view(B, N, num_heads, dh)
    q = apply_rope(rotary_pos, q)                                                     # :122
    k = apply_rope(rotary_pos, k)                                                     # :123
    q, k, v = (z.permute(0, 2, 1, 3) for z in (q, k, v))                              # :126-128
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dh)                     # :131-132
    attn = torch.softmax(scores, dim=-1)                                              # :133
    o = torch.matmul(attn, v)                                                         # :134
    x = o.permute(0, 2, 1, 3).reshape(B, N, d) + residual                             # :137-139
    if taps is not None:
        taps[prefix + "after_self"] = x

    # ---- cross-attention (:141-148) = nn.MultiheadAttention.forward, seq-first ----
    residual = x
    u = F.layer_norm(x, (d,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], LN_EPS)
    w, b = sd[prefix + "cross_attn.in_proj_weight"], sd[prefix + "cross_attn.in_proj_bias"]
    qc = F.linear(u, w[:d], b[:d])                       # functional.py _in_projection_packed (q is not k)
    kc = F.linear(text_emb, w[d:2 * d], b[d:2 * d])
    vc = F.linear(text_emb, w[2 * d:], b[2 * d:])
    qc = qc.view(B, N, num_heads, dh).permute(0, 2, 1, 3)
    kc = kc.view(B, T, num_heads, dh).permute(0, 2, 1, 3)
    vc = vc.view(B, T, num_heads, dh).permute(0, 2, 1, 3)
    qc = qc * math.sqrt(1.0 / float(dh))                 # functional.py: q_scaled = q * sqrt(1/E_head)
    ac = torch.softmax(torch.matmul(qc, kc.transpose(-2, -1)), dim=-1)   # no key-padding mask (SURVEY B-4)
    if drop_mask is not None:                            # train mode with a GIVEN keep-mask [B,H,N,T] (hash_dropout_mask)
        ac = ac * drop_mask * (1.0 / (1.0 - dropout_p))
    elif dropout_p > 0.0:
        ac = F.dropout(ac, p=dropout_p)                  # only in train mode (DiT.py:90-91)
    oc = torch.matmul(ac, vc).permute(0, 2, 1, 3).reshape(B, N, d)
    oc = F.linear(oc, sd[prefix + "cross_attn.out_proj.weight"], sd[prefix + "cross_attn.out_proj.bias"])
    x = oc + residual                                                                 # :148
    if taps is not None:
        taps[prefix + "after_cross"] = x

    # ---- gated MLP (:150-155); nn.GELU() default = exact erf form (:96) ----
    residual = x
    u = F.layer_norm(x, (d,), sd[prefix + "norm3.weight"], sd[prefix + "norm3.bias"], LN_EPS)
    h = F.gelu(F.linear(u, sd[prefix + "mlp_fc1.weight"], sd[prefix + "mlp_fc1.bias"]))
    g = torch.sigmoid(F.linear(u, sd[prefix + "gate.weight"], sd[prefix + "gate.bias"]))
    x = F.linear(h * g, sd[prefix + "mlp_fc2.weight"], sd[prefix + "mlp_fc2.bias"]) + residual
    if taps is not None:
        taps[prefix + "after_mlp"] = x
    return x


# --------------------------------------------------------------------------- #
# DiTTO.forward                                          src/model/DiTTO.py:66-94
# --------------------------------------------------------------------------- #
def time_embedding(sd: Mapping[str, Tensor], t: Tensor) -> Tensor:
    e = sd["t_embedding.weight"][t]                                                   # :75
    e = F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])               # :76
    return F.linear(F.silu(e), sd["time_embed.2.weight"], sd["time_embed.2.bias"])


def ditto_forward(sd: Mapping[str, Tensor], num_layers: int, num_heads: int, x: Tensor, text_emb: Tensor,
                  t: Tensor, taps: Optional[Dict[str, Tensor]] = None, dropout_p: float = 0.0,
                  dropout_seed: Optional[int] = None) -> Tensor:
    """dropout_p > 0 with dropout_seed: train-mode forward whose cross-attention keep-masks are the build's
    counter-based hash (hash_dropout_mask) instead of torch's Philox draw, so values can be compared."""
    temb = time_embedding(sd, t)                                                      # :75-76
    rotary_pos = rotary_table(sd["rotary.inv_freq"], x.shape[1])                      # :79-80
    x_skip = F.linear(x, sd["proj_in.weight"], sd["proj_in.bias"])                    # :83 (RAW input)
    h = global_adaln(sd, x, temb, text_emb)                                           # :86
    if taps is not None:
        taps["after_adaln"] = h
    for l in range(num_layers):                                                       # :89-90
        mask = None
        if dropout_p > 0.0 and dropout_seed is not None:
            mask = hash_dropout_mask(dropout_seed, l, x.shape[0], num_heads, x.shape[1], text_emb.shape[1], dropout_p)
        h = dit_block(sd, f"blocks.{l}.", h, text_emb, rotary_pos, num_heads, taps, dropout_p, mask)
    return x_skip + F.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])           # :93-94


# --------------------------------------------------------------------------- #
# schedule, q_sample                                    src/model/DiTTO.py:96-126
# --------------------------------------------------------------------------- #
def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> Tensor:
    """src/model/DiTTO.py:96-104 (returns CLIPPED BETAS, despite the variable names)."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps)
    alphas_cumprod = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0.0001, 0.9999)


def q_sample(alphas_cumprod_buffer: Tensor, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """src/model/DiTTO.py:106-126, bug-for-bug: the buffer holds betas."""
    t = t.long()
    a = alphas_cumprod_buffer[t] ** 0.5
    b = (1 - alphas_cumprod_buffer[t]) ** 0.5
    return a.reshape(-1, 1, 1) * x_start + b.reshape(-1, 1, 1) * noise


# --------------------------------------------------------------------------- #
# sampler                                   src/model/SpeechGenerator.py:70-72,130-164
# --------------------------------------------------------------------------- #
def sampler_tables(timesteps: int):
    """src/model/SpeechGenerator.py:70-72"""
    betas = cosine_beta_schedule(timesteps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    return betas, alphas, alphas_cumprod


def p_sample_update(x: Tensor, noise_pred: Tensor, t: Tensor, betas: Tensor, alphas: Tensor,
                    alphas_cumprod: Tensor, noise: Tensor) -> Tensor:
    """src/model/SpeechGenerator.py:137-145"""
    beta_t = betas[t].view(-1, 1, 1)
    alpha_t = alphas[t].view(-1, 1, 1)
    alpha_cumprod_t = alphas_cumprod[t].view(-1, 1, 1)
    mask = (t > 0).float().view(-1, 1, 1)
    return (1 / torch.sqrt(alpha_t)) * (
        x - (1 - alpha_t) / torch.sqrt(1 - alpha_cumprod_t) * noise_pred
    ) + mask * torch.sqrt(beta_t) * noise


def sample_latents(sd: Mapping[str, Tensor], num_layers: int, num_heads: int, x_init: Tensor, text_emb: Tensor,
                   timesteps: int, noises, keep=()):
    """src/model/SpeechGenerator.py:149-164 with the noise injected: `noises[i]` is the z of the
    i-th executed step (i = 0 is t = timesteps-1).  Returns (x_final, {i: x after step i})."""
    betas, alphas, ac = sampler_tables(timesteps)
    x = x_init
    kept = {}
    for i, t_val in enumerate(reversed(range(timesteps))):                            # :161
        t = torch.full((x.shape[0],), t_val, dtype=torch.long)                        # :162
        eps = ditto_forward(sd, num_layers, num_heads, x, text_emb, t)                # :135
        x = p_sample_update(x, eps, t, betas, alphas, ac, noises[i])                  # :137-145
        if i in keep:
            kept[i] = x.clone()
    return x, kept


# --------------------------------------------------------------------------- #
# either side of the loop
# --------------------------------------------------------------------------- #
def vq_indices(codebook: Tensor, latents: Tensor) -> Tensor:
    """VectorQuantizer.forward, src/components/VectorQuantizer.py:22-43 (latents [B,C,F,D] -> indices [B,C,F])."""
    Bz, C, Fr, D = latents.shape
    flat = latents.reshape(-1, D)
    distances = (torch.sum(flat ** 2, dim=1, keepdim=True)                    # :34-38
                 - 2 * torch.matmul(flat, codebook.T)
                 + torch.sum(codebook ** 2, dim=1))
    return torch.argmin(distances, dim=-1).view(Bz, C, Fr)                    # :40-41


def code_embed_mean(embedding_head: Tensor, codes: Tensor, max_length: int) -> Tensor:
    """EnCodec.forward's lookup (src/components/EnCodec.py:35-37) followed by the clip + mean over the codebooks
    of src/model/SpeechGenerator.py:97-98: codes [B,C,F] -> [B, min(F,max_length), D]."""
    return embedding_head[codes][:, :, :max_length].mean(dim=1)


def ddim_coefficients(alphas_cumprod: Tensor, t: int, t_prev: int, eta: float = 0.0):
    """One step of the strided sampler the paper uses (doc/1686_DiTTo_TTS_Diffusion_Trans.pdf App. A; Song et al.
    2021 eq. 12) written as x' = a*x + ce*eps + cz*z.  NOT in the reference: parity anchor is the
    published formula.  t_prev < 0 means the final step (abar_prev = 1)."""
    ab_t = alphas_cumprod[t]
    ab_p = alphas_cumprod[t_prev] if t_prev >= 0 else torch.tensor(1.0)
    sigma = eta * torch.sqrt((1 - ab_p) / (1 - ab_t)) * torch.sqrt(1 - ab_t / ab_p)
    a = torch.sqrt(ab_p / ab_t)
    ce = torch.sqrt(torch.clamp(1 - ab_p - sigma ** 2, min=0.0)) - torch.sqrt(ab_p * (1 - ab_t) / ab_t)
    return float(a), float(ce), float(sigma)


def sample_latents_strided(sd, num_layers, num_heads, x_init, text_emb, timesteps, n_steps, noises=None, eta=0.0,
                           cfg_scale=None, null_text=None):
    """DDIM-style loop over n_steps evenly spaced timesteps, optional classifier-free guidance
    eps = eps_u + w (eps_c - eps_u)."""
    _, _, ac = sampler_tables(timesteps)
    taus = strided_timesteps(timesteps, n_steps)
    x = x_init
    for i, t_val in enumerate(taus):
        t = torch.full((x.shape[0],), t_val, dtype=torch.long)
        eps = ditto_forward(sd, num_layers, num_heads, x, text_emb, t)
        if cfg_scale is not None:
            eps_u = ditto_forward(sd, num_layers, num_heads, x, null_text, t)
            eps = eps_u + cfg_scale * (eps - eps_u)
        t_prev = taus[i + 1] if i + 1 < len(taus) else -1
        a, ce, cz = ddim_coefficients(ac, t_val, t_prev, eta)
        z = noises[i] if (noises is not None and cz != 0.0) else torch.zeros_like(x)
        x = a * x + ce * eps + cz * z
    return x


def strided_timesteps(timesteps: int, n_steps: int):
    """n_steps descending timesteps T-1 ... spaced evenly (ends at the smallest multiple of the stride)."""
    stride = timesteps / n_steps
    return [int(round(timesteps - 1 - i * stride)) for i in range(n_steps)]


# --------------------------------------------------------------------------- #
# train-mode dropout mask of the build (NOT in the reference: torch draws its mask from Philox; the HIP path
# draws it from this counter-based hash so the backward can regenerate it — csrc/train.hip drop_stream/drop_keep)
# --------------------------------------------------------------------------- #
def _lowbias32(h):
    import numpy as np
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def _elem_hash(a, b, ctr):
    """per-element hash (csrc/common.h drop_keep): x = a + ctr; x ^= x >> 13; x += b; x ^= x >> 9; (x & 0xFFFFFF) * C — the two
    stream words enter by ADDs (non-linear over GF(2)), each in front of an xor-fold; the multiply is the 24-bit one the GPU
    runs at full rate"""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)
    x = (a.astype(np.uint64) + ctr.astype(np.uint64)) & M32
    x ^= x >> np.uint64(13)
    x = (x + b.astype(np.uint64)) & M32
    x ^= x >> np.uint64(9)
    return ((x & np.uint64(0xFFFFFF)) * np.uint64(0xD2B74F)) & M32


def hash_dropout_streams(seed: int, layer: int, nbh: int):
    """the two 32-bit words (a, b) of each (batch, head) stream bh = b * H + h (csrc/common.h drop_stream)"""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    bh = np.arange(nbh, dtype=np.uint64)
    inner = _lowbias32((hi + np.uint64(layer) * np.uint64(0x632BE5AB) + bh * np.uint64(0x9E3779B1)) & M32)
    a = _lowbias32(lo ^ inner)
    return a, _lowbias32(a ^ np.uint64(0x5BD1E995))


def hash_dropout_mask(seed: int, layer: int, B: int, H: int, Sq: int, Skv: int, p: float) -> Tensor:
    """keep-mask float32 [B, H, Sq, Skv] in {0, 1}: keep iff hash(stream(seed, layer, b*H+h), i, j) >= p * 2^32."""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)
    thr = np.uint64(min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF))
    a, b = hash_dropout_streams(seed, layer, B * H)                                   # [B*H] each
    i = np.arange(Sq, dtype=np.uint64)[:, None]
    j = np.arange(Skv, dtype=np.uint64)[None, :]
    ctr = (i * np.uint64(0x9E3779B1) + j * np.uint64(0x85EBCA6B)) & M32               # [Sq, Skv]
    h = _elem_hash(a[:, None, None], b[:, None, None], ctr[None])
    return torch.from_numpy((h >= thr).astype(np.float32)).view(B, H, Sq, Skv)


# ---------------------------------------------------------------------------------------------------------------
# Speech-length predictor: the decoder stack after the two pretrained encoders.
# Reference: src/model/SpeechLP.py:22-32 (construction: nn.TransformerDecoderLayer defaults = post-norm, ReLU,
# LayerNorm eps 1e-5, no final norm, dim_feedforward = d_model * nhead) and :47-55 (forward).  The layer math is
# torch's (third-party dependency, torch/nn/modules/transformer.py TransformerDecoderLayer.forward with
# norm_first=False; attention = F.multi_head_attention_forward) restated with plain ops.  Pinned by
# tests/golden/G7_slp*.npz, captured from the reference's own SLP.forward with the pretrained encoders replaced by
# pass-through stand-ins (tests/golden/make_golden.py make_slp), and against nn.TransformerDecoder live.
# ---------------------------------------------------------------------------------------------------------------
def _mha(sd: Mapping[str, Tensor], prefix: str, q_in: Tensor, kv_in: Tensor, nhead: int, causal: bool) -> Tensor:
    """nn.MultiheadAttention(batch_first=True) in eval mode; `causal` = the boolean triu(diagonal=1) tgt_mask of
    src/model/SpeechLP.py:57-61 (True = not allowed -> -inf before the softmax)."""
    d = q_in.shape[-1]
    W, b = sd[prefix + "in_proj_weight"].float(), sd[prefix + "in_proj_bias"].float()
    q = F.linear(q_in, W[:d], b[:d])
    k = F.linear(kv_in, W[d:2 * d], b[d:2 * d])
    v = F.linear(kv_in, W[2 * d:], b[2 * d:])
    B, S, _ = q.shape
    T = k.shape[1]
    dh = d // nhead
    q = q.view(B, S, nhead, dh).transpose(1, 2) * math.sqrt(1.0 / dh)
    k = k.view(B, T, nhead, dh).transpose(1, 2)
    v = v.view(B, T, nhead, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        mask = torch.triu(torch.ones(S, T), diagonal=1).bool()
        s = s.masked_fill(mask, float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, d)
    return F.linear(o, sd[prefix + "out_proj.weight"].float(), sd[prefix + "out_proj.bias"].float())


def slp_decode(sd: Mapping[str, Tensor], num_layers: int, nhead: int, z_text: Tensor, z_audio: Tensor):
    """src/model/SpeechLP.py:50-54: (z_text [B,T,d], z_audio [B,S,d]) -> (length logits [B,C], decoded [B,S,d])."""
    x, mem = z_audio.float(), z_text.float()
    d = x.shape[-1]
    for l in range(num_layers):
        p = f"transformer.layers.{l}."
        ln = lambda t, n: F.layer_norm(t, (d,), sd[p + n + ".weight"].float(), sd[p + n + ".bias"].float(), LN_EPS)  # noqa: E731
        x = ln(x + _mha(sd, p + "self_attn.", x, x, nhead, True), "norm1")
        x = ln(x + _mha(sd, p + "multihead_attn.", x, mem, nhead, False), "norm2")
        h = F.relu(F.linear(x, sd[p + "linear1.weight"].float(), sd[p + "linear1.bias"].float()))
        x = ln(x + F.linear(h, sd[p + "linear2.weight"].float(), sd[p + "linear2.bias"].float()), "norm3")
    logits = F.linear(x[:, -1, :], sd["length_predictor.weight"].float(), sd["length_predictor.bias"].float())
    return logits, x

"""The data-format steps either side of the denoise loop, on libditto_hip.so:
`VectorQuantizer` (reference src/components/VectorQuantizer.py, same constructor / parameter name / forward
contract), the two embedding lookups that build z_speech / z_text, and helpers for the strided (DDIM) sampler
with classifier-free guidance.  No CPU path."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip
from .engine import _stream


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: ditto_tts_amd runs only on an MI355X (no CPU / eager fallback)")


class VectorQuantizer(nn.Module):
    """Convert latents into discrete indices using a codebook — reference src/components/VectorQuantizer.py:4-43."""

    def __init__(self, codebook_size, latent_dim):
        super().__init__()
        self.codebook_size = codebook_size
        self.latent_dim = latent_dim
        self.codebook = nn.Parameter(torch.randn(codebook_size, latent_dim))
        nn.init.xavier_uniform_(self.codebook)

    def forward(self, latents):
        """latents [B, C, F, D] -> indices int64 [B, C, F] (nearest codebook row, squared L2, fp32)."""
        _need_cuda(latents, "latents")
        Bz, C, Fr, D = latents.shape
        x = latents.detach().float().contiguous()
        cb = self.codebook.detach().to(x.device).float().contiguous()
        idx = torch.empty(Bz * C * Fr, dtype=torch.int64, device=x.device)
        scratch = torch.empty(self.codebook_size, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().ditto_vq_argmin(x.data_ptr(), cb.data_ptr(), idx.data_ptr(), Bz * C * Fr,
                                            self.codebook_size, D, scratch.data_ptr(), _stream()))
        return idx.view(Bz, C, Fr)


def embedding_gather(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """nn.Embedding lookup (GPT-2 wte for z_text, reference src/model/SpeechGenerator.py:101-103)."""
    _need_cuda(table, "table")
    tab = table.detach().float().contiguous()
    flat = ids.to(tab.device).long().contiguous().view(-1)
    out = torch.empty(flat.numel(), tab.shape[1], dtype=torch.float32, device=tab.device)
    hip.check(hip.lib().ditto_embedding_gather(tab.data_ptr(), flat.data_ptr(), out.data_ptr(), flat.numel(),
                                               tab.shape[0], tab.shape[1], _stream()))
    return out.view(*ids.shape, tab.shape[1])


def code_embed_mean(embedding_head: torch.Tensor, codes: torch.Tensor, max_length: int) -> torch.Tensor:
    """EnCodec codes [B,C,F] -> z_speech [B, min(F,max_length), d]: embedding_head lookup, mean over the codebooks
    (reference src/components/EnCodec.py:35-37 + src/model/SpeechGenerator.py:97-98)."""
    _need_cuda(embedding_head, "embedding_head")
    tab = embedding_head.detach().float().contiguous()
    c = codes.to(tab.device).long().contiguous()
    Bz, C, Fr = c.shape
    Fout = min(Fr, max_length)
    out = torch.empty(Bz, Fout, tab.shape[1], dtype=torch.float32, device=tab.device)
    hip.check(hip.lib().ditto_code_embed_mean(tab.data_ptr(), c.data_ptr(), out.data_ptr(), Bz, C, Fr, Fout,
                                              tab.shape[0], tab.shape[1], _stream()))
    return out


def linear_update_(x, eps, noise, a, ce, cz):
    """x <- a[b]*x + ce[b]*eps + cz[b]*noise in place (a/ce/cz fp32 [B] device tensors)."""
    B = x.shape[0]
    hip.check(hip.lib().ditto_linear_update(x.data_ptr(), eps.data_ptr(), None if noise is None else noise.data_ptr(),
                                            a.data_ptr(), ce.data_ptr(), cz.data_ptr(), B, x.numel() // B, _stream()))
    return x


def cfg_combine(eps2, w: float):
    """eps2 [2B, ...] = [conditional; unconditional] -> eps_u + w*(eps_c - eps_u), [B, ...]."""
    B2 = eps2.shape[0]
    out = torch.empty((B2 // 2, *eps2.shape[1:]), dtype=torch.float32, device=eps2.device)
    hip.check(hip.lib().ditto_cfg_combine(eps2.data_ptr(), out.data_ptr(), float(w), out.numel(), _stream()))
    return out

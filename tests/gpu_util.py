"""Helpers shared by the -m gpu suites."""
import ctypes as C

import torch

from ditto_tts_amd import hip


def stream():
    return torch.cuda.current_stream().cuda_stream


def bf16(t):
    return t.to(torch.bfloat16).contiguous()


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b).clamp_min(1e-30))


def max_abs(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def asym(shape, seed, scale=1.0):
    """asymmetric, full-rank, sign-varying test data (guide: never check tile maps with symmetric data)"""
    from ditto_tts_amd.synth import hash_normal
    return hash_normal(shape, "asym", seed) * scale


def skip_unless_experimental(gemm_tile=0, attn_flags=0, fr_tile=0):
    """gemm_tile 130 (csrc/experimental/gemm_o3.hip), attn_flags bit 12 (attention_v4.hip) / bits 14, 15 (attention_w4.hip) and
    fr_tile 128 (gemm_fr128.hip) select opt-in A/B kernels that the default library is built without
    (DITTO_EXPERIMENTAL=1 python -m ditto_tts_amd.build --force)."""
    import pytest
    if (gemm_tile == 130 or (attn_flags & (4096 | 16384 | 32768)) or fr_tile == 128) and not hip.get_option("experimental"):
        pytest.skip("needs the csrc/experimental/ kernels (DITTO_EXPERIMENTAL=1 build)")


def experimental() -> bool:
    return bool(hip.get_option("experimental"))

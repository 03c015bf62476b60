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

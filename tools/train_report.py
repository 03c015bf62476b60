#!/usr/bin/env python3
"""Gradient parity report + training-step timing on one MI355X (supplementary to bench.py, which measures the
denoise step).  Prints worst/median per-tensor rel-L2 of the HIP gradients against fp32 autograd of the oracle at
a small shape, then times forward+backward+AdamW at the requested shape.

    python tools/train_report.py [--config C2] [--batch 8] [--seq 1024] [--steps 3] [--no-parity]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditto_tts_amd.config import PRESETS, DiTTOConfig  # noqa: E402
from ditto_tts_amd.modules import DiTTO  # noqa: E402
from ditto_tts_amd.synth import hash_normal, synthetic_inputs, synthetic_state_dict  # noqa: E402


def parity():
    from oracle import ditto_oracle as O
    cfg = DiTTOConfig(256, 3, 4, 256, 256, 50)
    B, N, T = 2, 128, 96
    x, text, t = synthetic_inputs(cfg, B, N, T, seed=7)
    target = hash_normal((B, N, 256), "noise", 9)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synthetic_state_dict(cfg, 4).items()}
    F.mse_loss(O.ditto_forward(sd, 3, 4, x, text, t), target).backward()
    m = DiTTO(256, 3, 4, 256, 256, 50)
    m.load_state_dict(synthetic_state_dict(cfg, 4))
    m = m.cuda().eval()
    F.mse_loss(m(x.cuda(), text.cuda(), t.cuda()), target.cuda()).backward()
    rows = []
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        a, b = p.grad.double().cpu().flatten(), sd[n].grad.double().flatten()
        rows.append((float((a - b).norm() / b.norm().clamp_min(1e-30)), n))
    rows.sort()
    print(f"gradient parity (3L d=256 h=4, B=2 N=128 T=96): median rel-L2 {rows[len(rows) // 2][0]:.2e}, "
          f"worst {rows[-1][0]:.2e} ({rows[-1][1]}), best {rows[0][0]:.2e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--text", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--wgrad-wgs", type=int, default=0)
    a = ap.parse_args()
    if a.wgrad_wgs:
        from ditto_tts_amd import hip
        hip.check(hip.lib().ditto_set_option(b"wgrad_wgs", a.wgrad_wgs))
    if not a.no_parity:
        parity()
    cfg = PRESETS[a.config]["cfg"]
    m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
    m.load_state_dict(synthetic_state_dict(cfg, 2))
    m = m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    B, N, T, d = a.batch, a.seq, a.text, cfg.hidden_dim
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, N, d, device="cuda", generator=g)
    text = torch.randn(B, T, d, device="cuda", generator=g)
    noise = torch.randn(B, N, d, device="cuda", generator=g)
    t = torch.randint(0, cfg.diffusion_steps, (B,), device="cuda", generator=g)
    times = {"fwd": [], "bwd": [], "opt": []}
    for i in range(a.steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = F.mse_loss(m(x, text, t), noise)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        opt.zero_grad(); loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if i:
            times["fwd"].append(t1 - t0); times["bwd"].append(t2 - t1); times["opt"].append(t3 - t2)
    med = {k: sorted(v)[len(v) // 2] * 1e3 for k, v in times.items()}
    fl = 3 * cfg.flops_per_utt_step(N, T, cached_kv=False) * B
    tot = sum(med.values())
    print(f"train step {a.config} B={B} N={N} T={T}: fwd {med['fwd']:.1f} ms, bwd {med['bwd']:.1f} ms, "
          f"AdamW+repack {med['opt']:.1f} ms  -> {B / tot * 1e3:.1f} utt/s, {fl / tot / 1e9:.0f} TFLOP/s "
          f"(3x forward FLOPs), loss {float(loss):.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()

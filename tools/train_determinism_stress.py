#!/usr/bin/env python3
"""The C2 training step N times with the same inputs, dropout seed and weights: every parameter gradient bitwise against the first
run's (no atomics anywhere in the backward; a race in a kernel's LDS ring would show up here as a flipped bit once in a while).
    python tools/train_determinism_stress.py [--reps 20] [--batch 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ditto_tts_amd.config import PRESETS
from ditto_tts_amd.modules import DiTTO
from ditto_tts_amd.synth import synthetic_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
p = PRESETS["C2"]
cfg, N, T, B = p["cfg"], p["N"], p["T"], a.batch
m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
m.load_state_dict(synthetic_state_dict(cfg, 2))
m = m.cuda().train()
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, N, cfg.hidden_dim, device="cuda", generator=g)
text = torch.randn(B, T, cfg.hidden_dim, device="cuda", generator=g)
noise = torch.randn(B, N, cfg.hidden_dim, device="cuda", generator=g)
t = torch.randint(0, cfg.diffusion_steps, (B,), device="cuda", generator=g)
ref, bad = None, 0
for r in range(a.reps):
    torch.manual_seed(123)                      # the dropout seed of the step
    m.zero_grad(set_to_none=True)
    junk = torch.empty((r % 5 + 1) * 7_000_000, device="cuda").normal_()   # perturb the allocator and the caches
    loss = F.mse_loss(m(x, text, t), noise)
    loss.backward()
    torch.cuda.synchronize()
    cur = {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}
    del junk
    if ref is None:
        ref = cur
        continue
    diff = [n for n in ref if not torch.equal(ref[n], cur[n])]
    if diff:
        bad += 1
        print(f"rep {r}: {len(diff)} tensors differ, e.g. {diff[:3]}")
print(f"{a.reps - 1 - bad} / {a.reps - 1} repeats bitwise equal to the first ({len(ref)} gradient tensors, loss {float(loss.detach()):.6f})")
sys.exit(1 if bad else 0)

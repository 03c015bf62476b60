#!/usr/bin/env python3
"""The C2 training step at batch sizes on both sides of every kernel-class boundary (low-latency, tiled, 64-row full-row, full-row),
round-4 forms (train_flags 0) against the round-3 forms (train_flags 30: two-launch gated backward, general gated epilogue, fp32 du,
fp32 tape): loss and worst parameter-gradient difference.  A robustness check of the dispatch predicates, not a parity test
(tests/test_gpu_train.py has those).    python tools/train_flags_sweep.py"""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from ditto_tts_amd import hip
from ditto_tts_amd.config import PRESETS
from ditto_tts_amd.modules import DiTTO
from ditto_tts_amd.synth import synthetic_state_dict, synthetic_inputs, hash_normal
cfg = PRESETS["C2"]["cfg"]
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
for B, N in ((12, 1024), (14, 1024), (16, 1024), (17, 1024), (5, 1000), (3, 1024), (1, 1024)):
    x, text, t = (z.cuda() for z in synthetic_inputs(cfg, B, N, 256, seed=3))
    target = hash_normal((B, N, cfg.hidden_dim), "noise", 4).cuda()
    res = {}
    for flag in (0, 30):
        hip.set_option("train_flags", flag)
        m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
        m.load_state_dict(synthetic_state_dict(cfg, 9)); m = m.cuda().eval()
        loss = F.mse_loss(m(x, text, t), target); loss.backward()
        res[flag] = (float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        del m
    hip.set_option("train_flags", 0)
    worst = max((rel(g, res[30][1][n]), n) for n, g in res[0][1].items())
    fin = all(torch.isfinite(g).all() for g in res[0][1].values())
    print(f"B={B} N={N}: loss {res[0][0]:.5f} / {res[30][0]:.5f}  worst grad rel {worst[0]:.2e} ({worst[1]})  finite {fin}", flush=True)

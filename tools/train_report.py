#!/usr/bin/env python3
"""Training-step timing on one MI355X (supplementary to bench.py, which measures the denoise step): forward +
backward + AdamW at the requested shape.  (Gradient parity against the oracle lives in tests/test_gpu_train.py; only
tests / smoke / bench's cpu_baseline may touch oracle/.)

    python tools/train_report.py [--config C2] [--batch 8] [--seq 1024] [--steps 3]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--text", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-parity", action="store_true", help="accepted and ignored (older command lines)")
    ap.add_argument("--wgrad-wgs", type=int, default=0)
    ap.add_argument("--train-flags", type=int, default=None, help="ditto_set_option('train_flags'): 1 = the rotation's "
                    "backward as its own pass (A/B)")
    ap.add_argument("--gemm-flags", type=int, default=None, help="ditto_set_option('gemm_flags'): 321 default; "
                    "+2048 = the two-buffer weight-gradient kernel (A/B)")
    a = ap.parse_args()
    if a.gemm_flags is not None:
        from ditto_tts_amd import hip as _hip
        _hip.check(_hip.lib().ditto_set_option(b"gemm_flags", a.gemm_flags))
    if a.train_flags is not None:
        from ditto_tts_amd import hip as _hip
        _hip.set_option("train_flags", a.train_flags)
    if a.wgrad_wgs:
        from ditto_tts_amd import hip
        hip.check(hip.lib().ditto_set_option(b"wgrad_wgs", a.wgrad_wgs))
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
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):                      # as a training loop runs it: no synchronisation between the phases
        loss = F.mse_loss(m(x, text, t), noise)
        opt.zero_grad(); loss.backward()
        opt.step()
    torch.cuda.synchronize()
    loop_ms = (time.perf_counter() - t0) / a.steps * 1e3
    fl = 3 * cfg.flops_per_utt_step(N, T, cached_kv=False) * B
    tot = sum(med.values())
    print(f"train step {a.config} B={B} N={N} T={T}: fwd {med['fwd']:.1f} ms, bwd {med['bwd']:.1f} ms, "
          f"AdamW+repack {med['opt']:.1f} ms (sum {tot:.1f}; back to back {loop_ms:.1f} ms/step)  -> {B / tot * 1e3:.1f} utt/s, {fl / tot / 1e9:.0f} TFLOP/s "
          f"(3x forward FLOPs), loss {float(loss):.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()

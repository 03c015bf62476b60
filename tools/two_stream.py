#!/usr/bin/env python3
"""Experiment: one batch of 32 utterances vs two half batches on two HIP streams (kernels of different phases —
HBM-bound epilogues / LayerNorms vs MFMA-bound main loops — may overlap across the halves)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ditto_tts_amd.config import PRESETS  # noqa: E402
from ditto_tts_amd.engine import DenoiseEngine  # noqa: E402
from ditto_tts_amd.sampler import SpeechGenerator  # noqa: E402
from ditto_tts_amd.modules import DiTTO  # noqa: E402
from ditto_tts_amd.synth import synthetic_state_dict  # noqa: E402

p = PRESETS["C2"]
cfg, N, T, B = p["cfg"], p["N"], p["T"], p["B"]
dev = torch.device("cuda")
sd = synthetic_state_dict(cfg, seed=1234)
m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
m.load_state_dict(sd)
m = m.to(dev).eval()
sg = SpeechGenerator(ditto_model=m, device=dev)
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engines = [DenoiseEngine(cfg, sd, dev) for _ in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
full = m.engine(dev)
text = torch.randn(B, T, cfg.text_dim, device=dev)
x = torch.randn(B, N, cfg.hidden_dim, device=dev)
z = torch.randn_like(x)
t = torch.full((B,), cfg.diffusion_steps - 1, device=dev, dtype=torch.long)
cond = full.prepare_text(text, N)
h = B // nsplit
xs = [x[i * h:(i + 1) * h].contiguous() for i in range(nsplit)]
zs = [z[i * h:(i + 1) * h].contiguous() for i in range(nsplit)]
ts = [t[i * h:(i + 1) * h].contiguous() for i in range(nsplit)]
conds = [engines[i].prepare_text(text[i * h:(i + 1) * h].contiguous(), N) for i in range(nsplit)]
torch.cuda.synchronize()


def one(steps):
    for _ in range(steps):
        full.p_sample_(x, cond, t, z, sg.betas, sg.alphas, sg.alphas_cumprod)


def split(steps):
    for _ in range(steps):
        for i in range(nsplit):
            with torch.cuda.stream(streams[i]):
                engines[i].p_sample_(xs[i], conds[i], ts[i], zs[i], sg.betas, sg.alphas, sg.alphas_cumprod)


with torch.no_grad():
    for fn in (one, split):
        fn(2)
    torch.cuda.synchronize()
    for r in range(3):
        for name, fn in (("one stream  B=32", one), (f"{nsplit} streams B={h}", split)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(5)
            torch.cuda.synchronize()
            print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/step")

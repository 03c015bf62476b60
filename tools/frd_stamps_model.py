#!/usr/bin/env python3
"""gemm_frd's phases INSIDE the model step (bf16 residual stream, operands as the neighbouring kernels left them in the caches):
s_memtime stamps of the last cross out-projection + norm3 (K = 768) or fc2 + next norm1 (K = 3072) of a C2 forward.
Diagnostic builds: tools/build_diag_one.sh libditto_diag_frdstamp768.so gemm_frd.hip -DDITTO_DIAG_FRD_STAMP -DDITTO_DIAG_FRD_STAMP_K=768
(and ..3072); DITTO_HIP_LIB=<that library> python tools/frd_stamps_model.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
from ditto_tts_amd.config import PRESETS
from ditto_tts_amd.modules import DiTTO
from ditto_tts_amd.synth import synthetic_state_dict
raw = C.CDLL(hip.LIB_PATH)
p = PRESETS["C2"]
cfg, N, T, B = p["cfg"], p["N"], p["T"], p["B"]
dev = torch.device("cuda")
m = DiTTO(cfg.hidden_dim, cfg.num_layers, cfg.num_heads, cfg.time_dim, cfg.text_dim, cfg.diffusion_steps)
m.load_state_dict(synthetic_state_dict(cfg, seed=1234))
m = m.to(dev).eval()
eng = m.engine(dev)
text = torch.randn(B, T, cfg.text_dim, device=dev)
x = torch.randn(B, N, cfg.hidden_dim, device=dev)
t = torch.full((B,), cfg.diffusion_steps - 1, device=dev, dtype=torch.long)
cond = eng.prepare_text(text, N)
for rep in range(3):
    out = eng.forward(x, cond, t)
torch.cuda.synchronize()
n = 256 * 4 * 8
buf = (C.c_ulonglong * n)()
assert raw.ditto_diag_frd_stamps(buf, n) == 0
recs = [buf[i * 8:i * 8 + 8] for i in range(256 * 4)]
recs = [r for r in recs if r[5] == 1]
avg = [sum(r[i] for r in recs) / len(recs) for i in range(5)]
names = ("prologue: A slab 0 + bias DMA, residual -> accumulators, barrier", "main loop", "LayerNorm statistics (two passes, LDS exchange)",
         "normalise + stage through LDS + store issue (h, u)", "store drain")
print(f"gemm_frd<LN, RES, bf16 stream> in the C2 forward (last launch of the stamped depth); {len(recs)} waves; ticks per wave:")
for nme, v in zip(names, avg):
    print(f"    {nme:70s} {v:9.0f}  {100 * v / sum(avg):5.1f} %")

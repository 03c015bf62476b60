#!/usr/bin/env python3
"""Roofline lines for the kernels either side of the denoise loop, one JSON object:
VQ argmin (fp32 MFMA: TFLOP/s against the 157 TF fp32 peak), the two gathers (HBM GB/s against 8 TB/s).
    python tools/around_bench.py [--batch 32] > profiles/rNN_around_roofline.json"""
import argparse, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd.around import VectorQuantizer, code_embed_mean, embedding_gather
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = "cuda"
B, N, D, K = a.batch, 1024, 768, 1024
def timeit(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)
out = {}
vq = VectorQuantizer(K, D).to(dev)
vq.codebook.data.normal_(0, 0.05)
lat = torch.randn(B, 2, N, D, device=dev) * 0.06            # SpeechGenerator.py:117: latents repeated over the 2 codebooks
t = timeit(lambda: vq(lat))
R = B * 2 * N
out["vq_argmin"] = {"rows": R, "codes": K, "dim": D, "seconds": t, "tflops": 2.0 * R * K * D / t / 1e12, "peak_tflops": 157.3,
                    "frac": 2.0 * R * K * D / t / 1e12 / 157.3, "bound": "fp32 MFMA (v_mfma_f32_32x32x2_f32)"}
V = 50257
table = torch.randn(V, D, device=dev)
ids = torch.randint(0, V, (B, N), device=dev)
t = timeit(lambda: embedding_gather(table, ids))
by = B * N * D * 4 * 2
out["embedding_gather"] = {"rows": B * N, "dim": D, "seconds": t, "gbs": by / t / 1e9, "peak_gbs": 8000.0, "frac": by / t / 1e9 / 8000.0,
                           "bytes": by, "bound": "hbm"}
codes = torch.randint(0, 1024, (B, 2, 1500), device=dev)
etab = torch.randn(1024, D, device=dev)
t = timeit(lambda: code_embed_mean(etab, codes, N))
by = B * N * D * 4 + 1024 * D * 4 + B * 2 * N * 8               # algorithmic: output written once, the 3 MB table and the used codes read once
out["code_embed_mean"] = {"frames": B * N, "dim": D, "seconds": t, "gbs": by / t / 1e9, "peak_gbs": 8000.0, "frac": by / t / 1e9 / 8000.0,
                          "bytes": by, "bound": "hbm (algorithmic bytes: the 3 MB table is read from L2 2 x frames times, from HBM once)"}
print(json.dumps(out))

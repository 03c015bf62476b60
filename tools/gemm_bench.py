#!/usr/bin/env python3
"""Per-shape timing of the GEMM family through the C-ABI (ditto_gemm_bf16), both tile structures, random data.
    python tools/gemm_bench.py [--m 32768] [--iters 30]
Interleaved rounds in ONE process (guide rule 24); prints median TFLOP/s per (shape, structure)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ditto_tts_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=32768)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--tiles", default="128,256", help="variants: tile[/flags], e.g. 128,256/0,256/1,256/3")
ap.add_argument("--shapes", default="qkv,dxd,gated,fc2,final")
ap.add_argument("--custom", default="", help="extra shapes 'M,N,K;M,N,K' (epilogue 0)")
a = ap.parse_args()
lib = hip.lib()
dev = "cuda"
M = a.m
d = 768
SHAPES = {  # name: (N, K, epilogue, ldo)
    "qkv": (3 * d, d, 0, 3 * d), "dxd": (d, d, 0, d), "dxd_res": (d, d, 1, d), "gated": (8 * d, d, 3, 4 * d),
    "fc2": (d, 4 * d, 1, d), "final": (d, 2 * d, 4, d),
}
st = torch.cuda.current_stream().cuda_stream
bufs = {}
MS = {}
names = [n for n in a.shapes.split(",") if n]
for c in [c for c in a.custom.split(";") if c]:
    cm, cn, ck = (int(v) for v in c.split(","))
    SHAPES[f"c{cm}x{cn}x{ck}"] = (cn, ck, 0, cn)
    MS[f"c{cm}x{cn}x{ck}"] = cm
    names.append(f"c{cm}x{cn}x{ck}")
for name in names:
    N, K, epi, ldo = SHAPES[name]
    M = MS.get(name, a.m)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) * 0.1
    out = torch.zeros(M, ldo, device=dev, dtype=torch.bfloat16 if epi in (0, 3) else torch.float32)
    bufs[name] = (A, W, bias, out)

def run(name):
    N, K, epi, ldo = SHAPES[name]
    A, W, bias, out = bufs[name]
    M = A.shape[0]
    hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr() if epi == 1 else None,
                                  out.data_ptr(), ldo, M, N, K, epi, st))

tiles = a.tiles.split(",")


def select(v):
    t, _, f = v.partition("/")
    hip.check(lib.ditto_set_option(b"gemm_tile", int(t)))
    hip.check(lib.ditto_set_option(b"gemm_flags", int(f) if f else 321))
res = {(n, t): [] for n in bufs for t in tiles}
for n in bufs:
    for t in tiles:
        select(t)
        run(n)
torch.cuda.synchronize()
for it in range(a.iters):
    for n in bufs:
        for t in tiles:
            select(t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(n)
            e1.record()
            e1.synchronize()
            res[(n, t)].append(e0.elapsed_time(e1))
for n in bufs:
    N, K, epi, _ = SHAPES[n]
    M = bufs[n][0].shape[0]
    fl = 2.0 * M * N * K
    print(n, f"M={M} N={N} K={K}", "  ".join(
        f"[{t}] {statistics.median(res[(n, t)]) * 1e3:7.1f} us {fl / statistics.median(res[(n, t)]) / 1e9:7.1f} TF"
        f" (min {min(res[(n, t)]) * 1e3:.1f})" for t in tiles), flush=True)

#!/usr/bin/env python3
"""Full-row GEMM + fused LayerNorm (ditto_gemm_ln_bf16) against the two launches it replaces (ditto_gemm_bf16 with the
in-place residual epilogue + ditto_layernorm_bf16): parity of both outputs and interleaved timing.
    python tools/fr_bench.py [--m 32768]"""
import argparse, math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
ap = argparse.ArgumentParser(); ap.add_argument("--m", type=int, default=32768); ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--nores", action="store_true", help="new kernel without the residual (timing of the accumulator init)")
ap.add_argument("--noln", action="store_true", help="new kernel without the LayerNorm output")
ap.add_argument("--rot", type=int, default=8, help="fr_rot option: K-loop rotation period in tiles (0 = off)")
ap.add_argument("--variants", default="128/0,130/0,64/0,64/1800", help="new-kernel variants fr_tile/fr_stagger (10 ns ticks)")
a = ap.parse_args()
lib = hip.lib(); hip.check(lib.ditto_set_option(b"fr_rot", a.rot)); st = torch.cuda.current_stream().cuda_stream
M, N = a.m, 768
torch.manual_seed(0)
for K in (768, 3072):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, N, device="cuda")
    g = 1 + 0.1 * torch.randn(N, device="cuda"); b = 0.1 * torch.randn(N, device="cuda")
    big = torch.empty(64 << 20, device="cuda")           # 256 MB: flush the Infinity Cache between timed launches
    def old():
        h = res.clone(); u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        def run():
            hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), h.data_ptr(), h.data_ptr(), N, M, N, K, 1, st))
            hip.check(lib.ditto_layernorm_bf16(h.data_ptr(), g.data_ptr(), b.data_ptr(), u.data_ptr(), M, N, st))
        return h, u, run
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()      # stage-major [K/16][N][16]
    def new():
        h = res.clone(); u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        def run():
            hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), None if a.nores else h.data_ptr(),
                                             h.data_ptr(), N, None if a.noln else g.data_ptr(), None if a.noln else b.data_ptr(),
                                             None if a.noln else u.data_ptr(), N, M, N, K, st))
        return h, u, run
    variants = [tuple(int(x) for x in v.split("/")) for v in a.variants.split(",")]
    def sel(v):
        hip.check(lib.ditto_set_option(b"fr_tile", v[0])); hip.check(lib.ditto_set_option(b"fr_stagger", v[1]))
    h0, u0, r0 = old(); r0(); torch.cuda.synchronize()
    want = res + A.float() @ W.float().T + bias
    wu = torch.nn.functional.layer_norm(want, (N,), g, b, 1e-5)
    h1, u1, r1 = new()
    ref = None
    for v in variants:
        sel(v); h1.copy_(res); r1(); torch.cuda.synchronize()
        if ref is None: ref = (h1.clone(), u1.clone())
        print(f"K={K} {v}: h old {float((h0-want).abs().max()):.2e} new {float((h1-want).abs().max()):.2e} | "
              f"u old {float((u0.float()-wu).abs().max()):.2e} new {float((u1.float()-wu).abs().max()):.2e} | "
              f"bitwise vs first variant: h {bool(torch.equal(h1, ref[0]))} u {bool(torch.equal(u1, ref[1]))}", flush=True)
    ts = {"old": []}; ts.update({v: [] for v in variants})
    for it in range(a.iters):
        for name in ts:
            h, u, run = (h0, u0, r0) if name == "old" else (h1, u1, r1)
            if name != "old": sel(name)
            h.copy_(res); big.zero_(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize()
            ts[name].append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * M * N * K
    for name in ts:
        med = statistics.median(ts[name])
        print(f"  {str(name):>12s}: {med:7.1f} us (min {min(ts[name]):.1f})  GEMM-only rate if the rest were free: {fl / med / 1e6:7.1f} TF")

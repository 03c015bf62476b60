#!/usr/bin/env python3
"""Bit-compare GEMM tile structures against the 128x128 kernel for every epilogue the C-ABI exposes (0 bias->bf16,
1 bias+residual->fp32 in place, 3 gated, 4 bias->fp32), production flags and slow-epilogue flags.
    python tools/gemm_check.py [--tiles 256,131,192]"""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
ap = argparse.ArgumentParser()
ap.add_argument("--tiles", default="256,131,192")
a = ap.parse_args()
lib = hip.lib()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
bad = 0
for (M, N, K) in [(256, 2304, 768), (512, 6144, 768), (1024, 768, 3072), (384, 768, 1536), (300, 768, 768), (2048, 2304, 256)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, N, device="cuda")
    for epi in (0, 1, 3, 4):
        def run(tile, flags):
            hip.check(lib.ditto_set_option(b"gemm_tile", tile)); hip.check(lib.ditto_set_option(b"gemm_flags", flags))
            ldo = N // 2 if epi == 3 else N
            out = res.clone() if epi == 1 else torch.zeros(M, ldo, device="cuda", dtype=torch.bfloat16 if epi in (0, 3) else torch.float32)
            hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), out.data_ptr() if epi == 1 else None,
                                          out.data_ptr(), ldo, M, N, K, epi, st))
            torch.cuda.synchronize()
            return out
        ref = run(128, 321)
        for t in [int(x) for x in a.tiles.split(",")]:
            for fl in (321, 1345, 16705, 17729):
                o = run(t, fl)
                d = (o.float() - ref.float()).abs().max().item()
                scale = ref.float().abs().max().item()
                ok = d <= 2e-2 * scale
                if not ok:
                    bad += 1
                print(f"M={M} N={N} K={K} epi={epi} tile={t} flags={fl}: max|d|={d:.3e} (scale {scale:.2f}) {'ok' if ok else 'MISMATCH'}")
hip.check(lib.ditto_set_option(b"gemm_tile", 0)); hip.check(lib.ditto_set_option(b"gemm_flags", 321))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Where a gemm256 tile's cycles go: s_memtime stamps of a DIAGNOSTIC build
(tools/build_diag.sh libditto_diag_g256stamp.so -DDITTO_DIAG_G256_STAMP; DITTO_HIP_LIB=...).  Per wave and tile: start wait +
barriers | main loop | end barrier + next prologue issue | epilogue (bias read, arithmetic, store issue, loop back)."""
import argparse, ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="gated,qkv,plain")
ap.add_argument("--flags", type=int, default=321)
a = ap.parse_args()
lib = hip.lib(); raw = C.CDLL(hip.LIB_PATH); st = torch.cuda.current_stream().cuda_stream
hip.check(lib.ditto_set_option(b"gemm_tile", 256)); hip.check(lib.ditto_set_option(b"gemm_flags", a.flags))
M, K = 32768, 768
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
for name in a.shapes.split(","):
    N, epi = {"gated": (6144, 3), "qkv": (2304, 0), "plain": (6144, 0)}[name]
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N // 2 if epi == 3 else N, device="cuda", dtype=torch.bfloat16)
    buf = (C.c_ulonglong * 16)()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K, W.data_ptr(), bias.data_ptr(), None, out.data_ptr(), out.shape[1], M, N, K, epi, st))
        e1.record(); torch.cuda.synchronize()
        assert raw.ditto_diag_g256_stamps(buf) == 0
        both = list(buf)
    for grp in (0, 1):   # waves 0-3 (wm = 0) | waves 4-7 (wm = 1: one barrier behind in the loop)
        s = both[grp * 8:grp * 8 + 8]; tiles, waves = max(s[4], 1), max(s[5], 1)
        print(f"{name}: N={N} epi={epi} flags={a.flags}, wave group {grp}: {e0.elapsed_time(e1) * 1e3:.1f} us, {tiles / waves:.1f} tiles per wave; per wave and tile (s_memtime ticks):")
        for n, x in zip(("start: tile top -> main loop", "  of which: top -> AT the first barrier (zeroing, waits)", "main loop", "end barrier (+ prologue issue)",
                         "epilogue body (to its last store issued)", "loop back to the next tile's top"), (s[0], s[7], s[1], s[2], s[3], s[6])):
            print(f"    {n:58s} {x / tiles:9.1f}")
        tot = s[0] + s[1] + s[2] + s[3] + s[6]
        print(f"    {'sum':58s} {tot / tiles:9.1f}   (kernel / tiles-per-wave = {e0.elapsed_time(e1) * 1e3 / (tiles / waves):.2f} us)")
hip.check(lib.ditto_set_option(b"gemm_tile", 0)); hip.check(lib.ditto_set_option(b"gemm_flags", 321))

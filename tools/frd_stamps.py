#!/usr/bin/env python3
"""Where a tile of the 128-row full-row GEMM + residual + LayerNorm kernel (csrc/gemm_frd.hip) spends its time: s_memtime stamps of a
DIAGNOSTIC build (tools/build_diag_one.sh libditto_diag_frdstamp.so gemm_frd.hip -DDITTO_DIAG_FRD_STAMP; DITTO_HIP_LIB=...).
M = 32768 (256 tiles: one per CU), N = 768, K = 768 (cross out-projection + norm3) and K = 3072 (fc2 + the next norm1); fp32 residual form
(ditto_gemm_ln_bf16; the model's bf16-stream form moves half the residual / h bytes)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); raw = C.CDLL(hip.LIB_PATH); st = torch.cuda.current_stream().cuda_stream
M, N = 32768, 768
hip.set_option("fr_tile", 130); hip.set_option("fr_rot", 8)
g = torch.Generator(device="cuda").manual_seed(2)
for K in (768, 3072):
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
    bias = torch.zeros(N, device="cuda"); gamma = torch.ones(N, device="cuda"); beta = torch.zeros(N, device="cuda")
    h = torch.randn(M, N, device="cuda", generator=g)
    u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        flush.fill_(rep)                     # push the operands out of the Infinity Cache, as the model's other kernels do
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr(), h.data_ptr(), N, gamma.data_ptr(),
                                         beta.data_ptr(), u.data_ptr(), N, M, N, K, st))
        e1.record(); torch.cuda.synchronize()
    n = 256 * 4 * 8
    buf = (C.c_ulonglong * n)()
    assert raw.ditto_diag_frd_stamps(buf, n) == 0
    recs = [buf[i * 8:i * 8 + 8] for i in range(256 * 4)]
    recs = [r for r in recs if r[5] == 1]
    avg = [sum(r[i] for r in recs) / len(recs) for i in range(5)]
    print(f"gemm_frd<LN, RES> K = {K} (stamp build): {e0.elapsed_time(e1) * 1e3:.1f} us; {len(recs)} waves; ticks per wave:")
    names = ("prologue: A slab 0 + bias DMA, residual -> accumulators, barrier", f"main loop ({K // 16} stages of 24 MFMAs)",
             "LayerNorm statistics (two passes over 384 accumulators, LDS exchange)", "normalise + stage through LDS + store issue (h, u)", "store drain")
    for nme, x in zip(names, avg):
        print(f"    {nme:75s} {x:9.0f}  {100 * x / sum(avg):5.1f} %")
hip.set_option("fr_tile", 0); hip.set_option("fr_rot", 1)

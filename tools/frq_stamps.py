#!/usr/bin/env python3
"""Stamp timeline of the out-projection form of csrc/gemm_lnq.hip (ditto_gemm_resln_bf16) and, beside it, of the fused norm2 + q-projection
at 8 waves — DIAGNOSTIC build (tools/build_diag_one.sh libditto_diag_lnqstamp.so gemm_lnq.hip -DDITTO_DIAG_LNQ_STAMP)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); raw = C.CDLL(hip.LIB_PATH); st = torch.cuda.current_stream().cuda_stream
M, d = 32768, 768
g = torch.Generator(device="cuda").manual_seed(2)
A = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
h = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(d, d, device="cuda", generator=g) / math.sqrt(d)).to(torch.bfloat16)
gamma = torch.ones(d, device="cuda"); beta = torch.zeros(d, device="cuda"); bias = torch.zeros(d, device="cuda")
u = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(d * d * 2, dtype=torch.uint8, device="cuda")
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def report(title, nwaves, names):
    n = 2048 * 4 * 8
    buf = (C.c_ulonglong * n)()
    assert raw.ditto_diag_lnq_stamps(buf, n) == 0
    recs = [buf[i * 8:i * 8 + 8] for i in range(nwaves)]
    recs = [r for r in recs if r[3] == 1]
    avg = [sum(r[i] for r in recs) / len(recs) for i in range(3)]
    init = sum(r[6] for r in recs) / len(recs)
    print(title, f"{len(recs)} waves")
    for nme, x in zip(names, avg):
        print(f"    {nme:72s} {x:9.0f} ticks  {100 * x / sum(avg):5.1f} %")
    print(f"    {'  of the second: accumulator init (MODE 1: residual + bias -> accumulators)':72s} {init:9.0f} ticks")


for rep in range(3):
    flush.fill_(rep)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(lib.ditto_gemm_resln_bf16(A.data_ptr(), d, W.data_ptr(), bias.data_ptr(), h.data_ptr(), d, gamma.data_ptr(), beta.data_ptr(),
                                        u.data_ptr(), d, M, 16, ws.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
report(f"out-projection + residual + LayerNorm (gemm_lnq MODE 1), launch incl. the repack of W {e0.elapsed_time(e1) * 1e3:.1f} us;", 512 * 8,
       ("A rows -> LDS (+ bias / gamma / beta DMA, barrier)", "accumulator init + W ring prologue + 48 stages of MFMAs", "epilogue: statistics, h' and u staged and stored, drained"))
hip.set_option("fr_rot", 16)
for rep in range(3):
    flush.fill_(rep)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 1, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(), u.data_ptr(), d,
                                      M, d, 32, ws.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
hip.set_option("fr_rot", 1)
report(f"norm2 + q-projection at {hip.get_option('lnq_waves')} waves, launch incl. the repack {e0.elapsed_time(e1) * 1e3:.1f} us;", 512 * 8,
       ("LayerNorm of the 64 rows -> LDS", "W ring prologue + 48 stages of MFMAs", "epilogue: acc -> LDS -> global, drained"))

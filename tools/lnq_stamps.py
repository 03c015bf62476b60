#!/usr/bin/env python3
"""Where a tile of the fused norm2 + q-projection kernel (csrc/gemm_lnq.hip) spends its time: s_memtime stamps of a DIAGNOSTIC build
(tools/build_diag_one.sh libditto_diag_lnqstamp.so gemm_lnq.hip -DDITTO_DIAG_LNQ_STAMP; DITTO_HIP_LIB=...).  C2 shape, M = 32768."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); raw = C.CDLL(hip.LIB_PATH); st = torch.cuda.current_stream().cuda_stream
M, d = 32768, 768
g = torch.Generator(device="cuda").manual_seed(2)
h = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(d, d, device="cuda", generator=g) / math.sqrt(d)).to(torch.bfloat16)
gamma = torch.ones(d, device="cuda"); beta = torch.zeros(d, device="cuda"); bias = torch.zeros(d, device="cuda")
out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(d * d * 2, dtype=torch.uint8, device="cuda")
hip.set_option("fr_rot", 16)
import sys as _s
hot = "--hot" in _s.argv          # the rows written just before by another kernel (a copy), as in the model; default: whatever the caches hold
for rep in range(4):
    if hot:
        h.copy_(h.clone())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(lib.ditto_gemm_lnq_bf16(h.data_ptr(), d, 1, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), d,
                                      M, d, 32, ws.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
n = 2048 * 4 * 8
buf = (C.c_ulonglong * n)()
assert raw.ditto_diag_lnq_stamps(buf, n) == 0
recs = [buf[i * 8:i * 8 + 8] for i in range(2048 * 4)]
recs = [r for r in recs if r[3] == 1]
t0 = min(r[4] for r in recs); t1 = max(r[5] for r in recs)
avg = [sum(r[i] for r in recs) / len(recs) for i in range(3)]
print(f"gemm_lnq (stamp build; launch includes the repack of W) {e0.elapsed_time(e1) * 1e3:.1f} us; {len(recs)} waves; first start -> last end {t1 - t0} ticks")
for nme, x in zip(("LayerNorm of the 64 rows -> LDS (+ bias DMA, barrier)", "W ring prologue + 48 stages of MFMAs", "epilogue: acc -> LDS -> global, stores drained"), avg):
    print(f"    {nme:58s} {x:9.0f} ticks  {100 * x / sum(avg):5.1f} %")
ld = sum(r[7] for r in recs) / len(recs); vd = sum(r[6] for r in recs) / len(recs)
print(f"    inside the LayerNorm phase: rows landed after {ld:.0f} ticks, normalised + written to the LDS after {vd:.0f}, workgroup barrier passed after {avg[0]:.0f}")
hip.set_option("fr_rot", 1)

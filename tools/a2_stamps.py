#!/usr/bin/env python3
"""Where a wave's 64-key tile of attn64v2 spends its cycles: s_memtime stamps of a DIAGNOSTIC build
(tools/build_diag_one.sh libditto_diag_a2stamp.so attention.hip -DDITTO_DIAG_A2_STAMP; DITTO_HIP_LIB=...).  C2 cross-attention shape."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); raw = C.CDLL(hip.LIB_PATH); st = torch.cuda.current_stream().cuda_stream
B, H, Sq, Skv, dh = 32, 12, 1024, 1024, 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(1)
q = (torch.randn(B * Sq, d, device="cuda", generator=g) * (1.4426950408889634 / math.sqrt(dh))).to(torch.bfloat16)
k = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty_like(q)
hip.check(lib.ditto_set_option(b"attn_flags", 16))
buf = (C.c_ulonglong * 8)()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq, Skv, dh,
                                       1.0 / math.sqrt(dh), None, 0, st))
    e1.record(); torch.cuda.synchronize()
    assert raw.ditto_diag_a2_stamps(buf) == 0
s = list(buf)
tiles = max(s[6], 1)
print(f"attn64v2 (stamp build) B={B} H={H} Sq={Sq} Skv={Skv}: {e0.elapsed_time(e1) * 1e3:.1f} us; {s[7]} waves, {tiles / max(s[7], 1):.1f} tiles per wave;"
      " s_memtime ticks per wave and tile:")
names = ("DMA issue + K fragment reads + 8 S MFMAs issued", "S back, row maximum, lane exchange, raise check", "32 exponentials + bf16 packing",
         "V fragment reads + 12 P.V / row-sum MFMAs issued", "vmcnt(0) + workgroup barrier", "loop back / prologue")
tot = sum(s[:6])
for n, x in zip(names, s[:6]):
    print(f"    {n:55s} {x / tiles:8.1f}  {100.0 * x / tot:5.1f} %")
print(f"    {'sum':55s} {tot / tiles:8.1f}   (x 3 waves per SIMD sharing it: {tot / tiles / 3:.0f} per wave-tile of SIMD time)")
hip.check(lib.ditto_set_option(b"attn_flags", 3))

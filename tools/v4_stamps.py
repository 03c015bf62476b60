#!/usr/bin/env python3
"""Where attn64v4's cycles go: s_memtime stamps of a DIAGNOSTIC build (tools/build_diag.sh libditto_diag_v4stamp.so
-DDITTO_DIAG_V4_STAMP; select it with DITTO_HIP_LIB=...).  Shares, not lengths: the stamps fence overlaps the real kernel has."""
import argparse, ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--H", type=int, default=12)
ap.add_argument("--Sq", type=int, default=1024); ap.add_argument("--Skv", type=int, default=1024)
ap.add_argument("--flags", type=int, default=16 + 4096)
a = ap.parse_args()
lib = hip.lib()
raw = C.CDLL(hip.LIB_PATH)
B, H, Sq, Skv, dh = a.B, a.H, a.Sq, a.Skv, 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(1)
q = (torch.randn(B * Sq, d, device="cuda", generator=g) * (1.4426950408889634 / 8)).to(torch.bfloat16)
k = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty_like(q)
st = torch.cuda.current_stream().cuda_stream
hip.check(lib.ditto_set_option(b"attn_flags", a.flags))
buf = (C.c_ulonglong * 16)()
for rep in range(3):
    for _ in range(1):
        hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq, Skv, dh,
                                           0.125, None, 0, st))
    torch.cuda.synchronize()
    assert raw.ditto_diag_v4_stamps(buf) == 0
    s = list(buf)
    waves, iters = max(s[7], 1), max(s[6], 1)
    names = ["prologue", "slots 0..19 (per steady iteration)", "slots 20..39 (per steady iteration)", "wait + barrier (per iteration)",
             "drain", "epilogue"]
    per = [s[0] / waves, s[1] / iters, s[2] / iters, s[3] / iters, s[4] / waves, s[5] / waves]
    print(f"rep {rep}: waves {waves}, steady iterations per wave {iters / waves:.1f}")
    for n, x in zip(names, per):
        print(f"   {n:42s} {x:9.0f} cycles")
    for n, i in (("prologue: setup + Q loads issued", 8), ("prologue: DMA issue", 9), ("prologue: first tiles landed + barrier", 10),
                 ("epilogue: arithmetic + store issue", 11), ("the whole tile loop", 12)):
        print(f"      {n:42s} {s[i] / waves:9.0f} cycles")
    nkt = Skv // 64
    tot = per[0] + (per[1] + per[2] + per[3]) * (nkt - 2) + per[4] + per[5]
    print(f"   ~ per workgroup: {tot:.0f} cycles + iterations 0 and {nkt - 1} (not stamped)")
hip.check(lib.ditto_set_option(b"attn_flags", 3))

#!/usr/bin/env python3
"""Attention forward microbench through the C-ABI, kernel variants interleaved in ONE process on ONE device.
    python tools/attn_bench.py [--variants 3,11,16] [--B 32 --H 12 --Sq 1024 --Skv 1024] [--rounds 5] [--iters 10]
variant = attn_flags (bit 4 = 16: q pre-scaled -> attn64v2; 16+32+x: older kernel x on pre-scaled q)"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ditto_tts_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="3,11,16")
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--H", type=int, default=12)
ap.add_argument("--Sq", type=int, default=1024)
ap.add_argument("--Skv", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
lib = hip.lib()
B, H, Sq, Skv, dh = a.B, a.H, a.Sq, a.Skv, 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(B * Sq, d, device="cuda", generator=g).to(torch.bfloat16)
k = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
qs = (q.float() * (1.4426950408889634 / math.sqrt(dh))).to(torch.bfloat16)
out = torch.empty_like(q)
scale = 1.0 / math.sqrt(dh)
st = torch.cuda.current_stream().cuda_stream
variants = [int(x) for x in a.variants.split(",")]
times = {x: [] for x in variants}


def run(flags, n):
    hip.check(lib.ditto_set_option(b"attn_flags", flags))
    qq = qs if flags & 16 else q
    for _ in range(n):
        hip.check(lib.ditto_attention_bf16(qq.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H,
                                           Sq, Skv, dh, scale, None, 0, st))


for x in variants:
    run(x, 3)
torch.cuda.synchronize()
for r in range(a.rounds):
    for x in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(x, a.iters)
        e1.record()
        e1.synchronize()
        times[x].append(e0.elapsed_time(e1) / a.iters * 1e3)
fl = 4.0 * B * H * Sq * Skv * dh
for x in variants:
    t = sorted(times[x])[len(times[x]) // 2]
    print(f"attn_flags {x:3d}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s")
hip.check(lib.ditto_set_option(b"attn_flags", 3))

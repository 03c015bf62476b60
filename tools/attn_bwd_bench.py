#!/usr/bin/env python3
"""Training attention microbench through the C-ABI: the fused head_dim-64 forward (with log-sum-exp) and the two-kernel
backward (csrc/attention_bwd.hip) at the C2 training shape, plus a spot check of dq / dk / dv against fp32 autograd on one
(batch, head) so that a fast wrong kernel cannot pass for a result.
    python tools/attn_bwd_bench.py [--B 32 --H 12 --Sq 1024 --Skv 1024] [--p 0.0,0.1] [--rounds 5] [--iters 5]
A/B two builds on the same box:  DITTO_HIP_LIB=/path/to/libditto_base.so python tools/attn_bwd_bench.py ...
Per-kernel times: run it under  rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/attn_bwd_bench.py"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ditto_tts_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--H", type=int, default=12)
ap.add_argument("--Sq", type=int, default=1024)
ap.add_argument("--Skv", type=int, default=1024)
ap.add_argument("--p", default="0.0,0.1")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--no-check", action="store_true", help="skip the autograd spot check (knock-out builds are wrong by design)")
a = ap.parse_args()
lib = hip.lib()
B, H, Sq, Skv, dh = a.B, a.H, a.Sq, a.Skv, 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(1)
q, do = (torch.randn(B, Sq, d, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2))
k, v = (torch.randn(B, Skv, d, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2))
o = torch.empty_like(q)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
lse = torch.empty(B, H, Sq, dtype=torch.float32, device="cuda")
scale = 1.0 / math.sqrt(dh)
st = torch.cuda.current_stream().cuda_stream
nb = lib.ditto_attention_bwd_workspace_bytes(B, H, Sq, Skv, dh)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
seed, layer = 0x1234567890ABCDEF, 3


def fwd(p):
    hip.check(lib.ditto_attention_dropout_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, o.data_ptr(), d,
                                               lse.data_ptr(), B, H, Sq, Skv, dh, scale, p, seed, layer, ws.data_ptr(), nb, st))


def bwd(p):
    hip.check(lib.ditto_attention_bwd_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, do.data_ptr(), d, o.data_ptr(),
                                           d, lse.data_ptr(), dq.data_ptr(), d, dk.data_ptr(), d, dv.data_ptr(), d, B, H, Sq,
                                           Skv, dh, scale, p, seed, layer, ws.data_ptr(), nb, st))


def timed(fn, p):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn(p)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


# spot check (p = 0): fp32 autograd on the same bf16 operands, last batch entry, last head
fwd(0.0)
bwd(0.0)
torch.cuda.synchronize()
bi, hi = B - 1, H - 1
sl = slice(hi * dh, (hi + 1) * dh)
qf, kf, vf = (z[bi, :, sl].float().requires_grad_(True) for z in (q, k, v))
pr = torch.softmax(qf @ kf.T * scale, dim=-1)
(pr @ vf).backward(do[bi, :, sl].float())
for name, got, want in (() if a.no_check else (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad))):
    r = float((got[bi, :, sl].float() - want).norm() / want.norm())
    print(f"check {name}: rel-L2 {r:.3e}", "ok" if r < 2e-2 else "FAIL")
    if not r < 2e-2:
        sys.exit(1)

fl_f = 4.0 * B * H * Sq * Skv * dh
for p in [float(x) for x in a.p.split(",")]:
    tf, tb = [], []
    for _ in range(a.rounds):
        tf.append(timed(fwd, p))
        tb.append(timed(bwd, p))
    f, b = sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2]
    print(f"p={p}: forward+lse {f:.1f} us ({fl_f / f / 1e6:.0f} TFLOP/s)   backward (dq [+ delta] + dk,dv) {b:.1f} us "
          f"({3.5 * fl_f / b / 1e6:.0f} TFLOP/s on the 14 B H Sq Skv dh count, {2.5 * fl_f / b / 1e6:.0f} on the minimal 10)")

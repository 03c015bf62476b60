#!/usr/bin/env python3
"""attn64v4 against attn64v2 on crafted inputs (debug aid): where do they differ?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib()
st = torch.cuda.current_stream().cuda_stream
dh = 64


def run(flags, q, k, v, B, H, Sq, Skv):
    d = H * dh
    out = torch.full((B * Sq, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    hip.check(lib.ditto_set_option(b"attn_flags", flags))
    hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq, Skv, dh,
                                       1.0 / math.sqrt(dh), None, 0, st))
    torch.cuda.synchronize()
    hip.check(lib.ditto_set_option(b"attn_flags", 3))
    return out.float()


def report(name, a, b, Sq):
    dlt = (a - b).abs()
    print(f"{name}: max diff {float(dlt.max()):.4g}  (nan in v4: {int(torch.isnan(a).sum())})")
    if float(dlt.max()) > 0:
        rows = dlt.max(dim=1).values.view(-1, 32).max(dim=1).values
        cols = dlt.max(dim=0).values.view(-1, 8).max(dim=1).values
        print("   per 32-row block:", [f"{float(x):.3g}" for x in rows[:16]])
        print("   per 8-col block :", [f"{float(x):.3g}" for x in cols[:8]])


g = torch.Generator(device="cuda").manual_seed(3)
for (B, H, Sq, Skv) in [(1, 1, 256, 64), (1, 1, 256, 128), (1, 1, 256, 256), (1, 2, 512, 1024)]:
    d = H * dh
    print(f"== B={B} H={H} Sq={Sq} Skv={Skv}")
    q = (torch.randn(B * Sq, d, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    k = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
    zk = torch.zeros_like(k)
    ones_v = torch.ones_like(v)
    report("K=0, V=1 (expect 1)", run(16 + 4096, q, zk, ones_v, B, H, Sq, Skv), torch.ones(B * Sq, d, device="cuda"), Sq)
    report("K=0 (O = mean V)   ", run(16 + 4096, q, zk, v, B, H, Sq, Skv), run(16 + 256, q, zk, v, B, H, Sq, Skv), Sq)
    report("V=1 (expect 1)     ", run(16 + 4096, q, k, ones_v, B, H, Sq, Skv), torch.ones(B * Sq, d, device="cuda"), Sq)
    report("random             ", run(16 + 4096, q, k, v, B, H, Sq, Skv), run(16 + 256, q, k, v, B, H, Sq, Skv), Sq)

print("== probes: B=1 H=1 Sq=256 Skv=64, K = 0")
B, H, Sq, Skv = 1, 1, 256, 64
q = torch.zeros(Sq, 64, device="cuda", dtype=torch.bfloat16)
zk = torch.zeros(Skv, 64, device="cuda", dtype=torch.bfloat16)
for name, v in (("V[key, col] = key", torch.arange(Skv, device="cuda").float()[:, None].expand(Skv, 64)),
                ("V[key, col] = col", torch.arange(64, device="cuda").float()[None, :].expand(Skv, 64))):
    o4 = run(16 + 4096, q, zk, v.to(torch.bfloat16).contiguous(), B, H, Sq, Skv)
    o2 = run(16 + 256, q, zk, v.to(torch.bfloat16).contiguous(), B, H, Sq, Skv)
    print(name, "\n  v4 row 0 :", [f"{float(x):.4g}" for x in o4[0, :16]], "\n  v4 row 5 :", [f"{float(x):.4g}" for x in o4[5, :16]],
          "\n  v4 row 32:", [f"{float(x):.4g}" for x in o4[32, :16]], "\n  v2 row 0 :", [f"{float(x):.4g}" for x in o2[0, :16]])
# one-hot keys: V = e_key (64 x 64 identity): O[row, col] = P[row, key = col] / sum -> shows which keys are weighted wrongly
v = torch.eye(64, device="cuda").to(torch.bfloat16)
o4 = run(16 + 4096, q, zk, v, B, H, Sq, Skv)
print("V = I: v4 row 0 x 64 (expect 1/64 = 0.0156 everywhere):", [f"{float(x) * 64:.3g}" for x in o4[0]])
print("V = I: v4 row 40 x 64:", [f"{float(x) * 64:.3g}" for x in o4[40]])

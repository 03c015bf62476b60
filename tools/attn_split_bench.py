#!/usr/bin/env python3
"""The fused head_dim-64 attention at the reference's call pattern (B = 1, 12 heads, N = T = 1024) unsplit and split over the keys
("ll_mask" bit 2): per-launch HIP-event time; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); st = torch.cuda.current_stream().cuda_stream
B, H, Sq, Skv, dh = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 12, 1024, 1024, 64)))
d = H * dh
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.randn(B * n, d, device="cuda", generator=g).to(torch.bfloat16) for n in (Sq, Skv, Skv))
out = torch.empty(B * Sq, d, dtype=torch.bfloat16, device="cuda")
nws = lib.ditto_attention_workspace_bytes(B, H, Sq, Skv, dh)
ws = torch.empty(max(nws, 16), dtype=torch.uint8, device="cuda")
hip.check(lib.ditto_set_option(b"attn_flags", 16))
for mask in (3, 7, 3, 7):
    hip.set_option("ll_mask", mask)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, out.data_ptr(), d, B, H, Sq, Skv, dh,
                                               1 / math.sqrt(dh), ws.data_ptr(), ws.numel(), st))
        e1.record(); torch.cuda.synchronize()
    print(f"ll_mask {mask}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per attention call (B={B} H={H} Sq={Sq} Skv={Skv})")
hip.set_option("ll_mask", 7); hip.check(lib.ditto_set_option(b"attn_flags", 3))

#!/usr/bin/env python3
"""Does the ROW PITCH of the A operand matter?  Same GEMM (ditto_gemm_bf16 / ditto_gemm_ln_bf16), A rows K elements long at pitch
K (what the model has: fc2 reads the gated activation at pitch 4 d = 6 / 8 KiB) against pitch K + pad.  256 rows of one K-tile are
256 lines exactly one pitch apart: a power-of-two-ish pitch puts them on few L2 / memory channels.  Interleaved rounds, one process."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(3)
flush = torch.empty(600 << 20, dtype=torch.uint8, device="cuda")


def run(kind, M, N, K, pads, tile=0, rounds=5):
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    res = {}
    bufs = {}
    for pad in pads:
        buf = torch.empty(M, K + pad, device="cuda", dtype=torch.bfloat16)
        buf[:, :K] = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        bufs[pad] = buf
    h = torch.randn(M, N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda")
    u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    gamma = torch.ones(N, device="cuda"); beta = torch.zeros(N, device="cuda")
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
    hip.set_option("gemm_tile", tile)
    for r in range(rounds):
        for pad in pads:
            A = bufs[pad]
            flush.fill_(r)      # the operands come from HBM / Infinity Cache as after the model's other kernels
            A[:, :K].mul_(1.0)  # ... but A was just written (by the gated GEMM in the model)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if kind == "gemm":
                hip.check(lib.ditto_gemm_bf16(A.data_ptr(), K + pad, W.data_ptr(), bias.data_ptr(), h.data_ptr(), out.data_ptr(), N, M, N, K, 1, st))
            else:
                hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K + pad, Wp.data_ptr(), bias.data_ptr(), h.data_ptr(), out.data_ptr(), N,
                                                 gamma.data_ptr(), beta.data_ptr(), u.data_ptr(), N, M, N, K, st))
            e1.record(); torch.cuda.synchronize()
            if r: res.setdefault(pad, []).append(e0.elapsed_time(e1) * 1e3)
    hip.set_option("gemm_tile", 0)
    fl = 2.0 * M * N * K
    print(f"{kind:8s} M={M} N={N} K={K} tile={tile}: " + "   ".join(
        f"pitch K+{pad}: {sorted(v)[len(v) // 2]:7.1f} us ({fl / sorted(v)[len(v) // 2] / 1e6:5.0f} TF)" for pad, v in res.items()))


pads = (0, 64, 128, 32)
run("gemm", 16384, 1024, 4096, pads, tile=256)      # C5 fc2 on one round of 256 x 256 tiles
run("gemm", 16384, 1024, 4096, pads, tile=0)
run("gemm", 16384, 1024, 1024, pads, tile=256)      # C5 out-projection
run("gemm", 32768, 768, 3072, pads, tile=0)         # C2 fc2, tiled
run("gemm_ln", 32768, 768, 3072, pads)              # C2 fc2 + norm1 on the full-row kernel
run("gemm_ln", 32768, 768, 768, pads)               # C2 out-projection + norm3
run("gemm", 32768, 2304, 768, pads, tile=0)         # C2 QKV-like (plain epilogue 1 for the comparison)
run("gemm", 16384, 3072, 1024, pads, tile=0)        # C5 QKV-like

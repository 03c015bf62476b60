#!/usr/bin/env python3
"""Where does gemm_fr64 differ from gemm_fr?  Error maps per (row block, column block) on small shapes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
lib = hip.lib(); st = torch.cuda.current_stream().cuda_stream
N = 768
for (M, K, rot, ln, res, mode) in [(128, 64, 0, False, False, "rand"), (128, 64, 0, False, True, "rand"), (128, 128, 0, False, False, "rand"),
                                    (128, 768, 0, False, False, "rand"), (128, 768, 0, True, True, "rand"), (1024, 768, 8, True, True, "rand"),
                                    (128, 256, 0, False, False, "kid")]:
    torch.manual_seed(0)
    hip.check(lib.ditto_set_option(b"fr_rot", rot))
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    if mode == "kid":      # A[m, k] = 1 if k == k0 else 0, W[n, k] = k: output = k0 where products are picked up
        A = torch.zeros(M, K, device="cuda"); W = torch.arange(K, device="cuda", dtype=torch.float32).expand(N, K).contiguous()
        A = torch.ones(M, K, device="cuda").to(torch.bfloat16); W = W.to(torch.bfloat16)
    Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
    bias = torch.randn(N, device="cuda") * 0.1
    r0 = torch.randn(M, N, device="cuda")
    g = 1 + 0.1 * torch.randn(N, device="cuda"); b = 0.1 * torch.randn(N, device="cuda")
    outs = {}
    for tile in (128, 64):
        hip.check(lib.ditto_set_option(b"fr_tile", tile)); hip.check(lib.ditto_set_option(b"fr_stagger", 0))
        h = r0.clone() if res else torch.zeros(M, N, device="cuda")
        u = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr() if res else None, h.data_ptr(), N,
                                         g.data_ptr() if ln else None, b.data_ptr() if ln else None, u.data_ptr() if ln else None, N, M, N, K, st))
        torch.cuda.synchronize(); outs[tile] = (h, u)
    want = A.float() @ W.float().T + bias + (r0 if res else 0)
    e128 = float((outs[128][0] - want).abs().max()); d = (outs[64][0] - want).abs()
    print(f"M={M} K={K} rot={rot} ln={ln} res={res} {mode}: err128 {e128:.2e} err64 {float(d.max()):.2e} bitwise h {torch.equal(outs[64][0], outs[128][0])} u {torch.equal(outs[64][1], outs[128][1])}")
    if float(d.max()) > 1e-3:
        blk = d[:128].view(4, 32, 24, 32).amax(dim=(1, 3))     # [row block of 32][col block of 32]
        print("  max err per (32-row block, 32-col block) of the first 128 rows:")
        for r in range(4): print("   ", " ".join(f"{float(x):6.2f}" for x in blk[r]))
        sub = d[:32, :32]
        print("  rows with error in block (0,0):", (sub.amax(1) > 1e-3).nonzero().flatten().tolist())
        print("  cols with error in block (0,0):", (sub.amax(0) > 1e-3).nonzero().flatten().tolist())
        if mode == "kid":
            print("  h64[0, :8] =", outs[64][0][0, :8].tolist(), " want", want[0, :4].tolist())

print("---- u mismatch structure (M = 4096, K = 768, random) ----")
torch.manual_seed(1)
M, K = 4096, 768
hip.check(lib.ditto_set_option(b"fr_rot", 0))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
bias = torch.randn(N, device="cuda") * 0.1
r0 = torch.randn(M, N, device="cuda")
g = 1 + 0.1 * torch.randn(N, device="cuda"); b = 0.1 * torch.randn(N, device="cuda")
outs = {}
for tile in (128, 64):
    hip.check(lib.ditto_set_option(b"fr_tile", tile)); hip.check(lib.ditto_set_option(b"fr_stagger", 0))
    h = r0.clone(); u = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr(), h.data_ptr(), N, g.data_ptr(), b.data_ptr(), u.data_ptr(), N, M, N, K, st))
    torch.cuda.synchronize(); outs[tile] = (h, u)
du = (outs[64][1].float() - outs[128][1].float()).abs()
bad = du > 0
print("h equal:", torch.equal(outs[64][0], outs[128][0]), " u differing elements:", int(bad.sum()), "of", M * N, " max diff", float(du.max()))
rows = bad.any(1).nonzero().flatten()
print("rows with a difference:", len(rows), rows[:20].tolist())
if len(rows):
    r = int(rows[0]); cols = bad[r].nonzero().flatten()
    print(f"row {r}: {len(cols)} differing cols, first {cols[:16].tolist()}")
    print("  u64 ", outs[64][1][r, cols[:6]].tolist()); print("  u128", outs[128][1][r, cols[:6]].tolist())
    percol = bad.sum(0); print("columns by 32-block count of diffs:", percol.view(24, 32).sum(1).tolist())
    print("rows by r%64 // 32:", [(int((rows % 64 < 32).sum())), int((rows % 64 >= 32).sum())])

print("---- the failing pytest case: asym data, M = 128, K = 768 ----")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import asym
M, K = 128, 768
A = asym((M, K), 14).cuda().to(torch.bfloat16)
W = (asym((N, K), 15) / math.sqrt(K)).cuda().to(torch.bfloat16)
Wp = W.view(N, K // 16, 16).permute(1, 0, 2).contiguous()
bias = (0.1 * asym((N,), 16)).cuda(); r0 = asym((M, N), 17).cuda()
g = (1 + 0.1 * asym((N,), 18)).cuda(); b = (0.1 * asym((N,), 19)).cuda()
outs = {}
for tile in (128, 64):
    hip.check(lib.ditto_set_option(b"fr_tile", tile)); hip.check(lib.ditto_set_option(b"fr_stagger", 0))
    h = r0.clone(); u = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    hip.check(lib.ditto_gemm_ln_bf16(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), h.data_ptr(), h.data_ptr(), N, g.data_ptr(), b.data_ptr(), u.data_ptr(), N, M, N, K, st))
    torch.cuda.synchronize(); outs[tile] = (h, u)
du = (outs[64][1].float() - outs[128][1].float()).abs()
bad = du > 0
print("h equal:", torch.equal(outs[64][0], outs[128][0]), " u differing elements:", int(bad.sum()), "of", M * N, " max diff", float(du.max()))
rows = bad.any(1).nonzero().flatten()
print("rows with a difference:", len(rows), rows[:40].tolist())
if len(rows):
    r = int(rows[0]); cols = bad[r].nonzero().flatten()
    print(f"row {r}: {len(cols)} differing cols, first {cols[:16].tolist()}")
    print("  u64 ", outs[64][1][r, cols[:6]].tolist()); print("  u128", outs[128][1][r, cols[:6]].tolist())
    hrow = outs[128][0][r].double(); mu = hrow.mean(); var = ((hrow - mu) ** 2).mean()
    print("  exact mean", float(mu), "rstd", float(1 / torch.sqrt(var + 1e-5)))
    print("per 32-col block diffs:", bad.sum(0).view(24, 32).sum(1).tolist())

#!/usr/bin/env python3
"""Weight-gradient (K-major x K-major) GEMM microbench through the C-ABI (ditto_gemm_tn_bf16): every wgrad shape of a C2
training step, 128x128 against 256x256 tiles over a range of split-K factors, interleaved in one process.
    python tools/tn_bench.py [--rows 32768] [--iters 10]"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ditto_tts_amd import hip
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=32768); ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--splits128", default="1,2,3,4,6,8,11,14"); ap.add_argument("--splits256", default="1,2,3,4,6,8,12,16,24,28")
a = ap.parse_args()
lib = hip.lib(); st = torch.cuda.current_stream().cuda_stream
d = 768
SHAPES = {"dxd": (d, d), "kv 2d x d": (2 * d, d), "qkv 3d x d": (3 * d, d), "fc2 d x 4d": (d, 4 * d), "fc1|gate 8d x d": (8 * d, d)}
K = a.rows
for name, (Mo, No) in SHAPES.items():
    X = torch.randn(K, Mo, device="cuda").to(torch.bfloat16); Y = torch.randn(K, No, device="cuda").to(torch.bfloat16)
    out = torch.empty(Mo, No, device="cuda")
    ws = torch.empty(256 + 32 * Mo * No * 4, dtype=torch.uint8, device="cuda")
    res = []
    for tile, sl in ((128, a.splits128), (256, a.splits256)):
        for S in [int(v) for v in sl.split(",")]:
            ts = []
            for it in range(a.iters + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                hip.check(lib.ditto_gemm_tn_bf16(X.data_ptr(), Mo, Y.data_ptr(), No, out.data_ptr(), No, Mo, No, K, S, tile,
                                                 ws.data_ptr(), ws.numel(), st))
                e1.record(); e1.synchronize()
                if it >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
            res.append((statistics.median(ts), tile, S))
    fl = 2.0 * Mo * No * K
    best = {t: min(r for r in res if r[1] == t) for t in (128, 256)}
    print(f"{name:16s} 128x128 best S={best[128][2]:2d} {best[128][0]:7.1f} us ({fl/best[128][0]/1e6:6.0f} TF) | "
          f"256x256 best S={best[256][2]:2d} {best[256][0]:7.1f} us ({fl/best[256][0]/1e6:6.0f} TF)   all256: " +
          " ".join(f"{S}:{t:.0f}" for t, tl, S in res if tl == 256), flush=True)

#!/usr/bin/env python3
"""The fused norm2 + q-projection kernel (csrc/gemm_lnq.hip) in isolation on RANDOM data at the C2 shape (M = 32768 rows), both
MFMA shapes and both ring depths, interleaved rounds in one process (the 32x32x16 / 16x16x32 A/B on a real
kernel of the path).  Wall time per launch from HIP events; under `rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python3
tools/lnq_bench.py` the same launches give wave-cycles and the effective clock per variant (tools/pmc_kernels.py sums them up).
    python tools/lnq_bench.py [--rows 32768] [--rounds 5] [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ditto_tts_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=32768)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
lib = hip.lib()
dev = torch.device("cuda")
d, M = 768, a.rows
g = torch.Generator(device=dev).manual_seed(1)
h = torch.randn(M, d, device=dev, generator=g) * 1.5 + 0.3
hb = h.to(torch.bfloat16)
gamma = 1 + 0.1 * torch.randn(d, device=dev, generator=g)
beta = 0.1 * torch.randn(d, device=dev, generator=g)
W = (torch.randn(d, d, device=dev, generator=g) / d ** 0.5).to(torch.bfloat16)
bias = 0.1 * torch.randn(d, device=dev, generator=g)
scratch = torch.empty(d * d * 2, dtype=torch.uint8, device=dev)
out = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
stream = torch.cuda.current_stream().cuda_stream
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)       # between launches: evict h from the Infinity Cache
variants = [(32, 4), (32, 8), (16, 2), (16, 4)]
hip.set_option("fr_rot", 16)                                          # rotate as the model does (16 tiles per utterance)


def launch(shape, ring, src, is_bf16):
    hip.set_option("lnq_ring", ring)
    hip.check(lib.ditto_gemm_lnq_bf16(src.data_ptr(), d, int(is_bf16), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(),
                                      bias.data_ptr(), out.data_ptr(), d, M, d, shape, scratch.data_ptr(), stream))


times = {(v, s): [] for v in variants for s in ("fp32", "bf16")}
for _ in range(3):
    for v in variants:
        launch(*v, hb, True)
torch.cuda.synchronize()
for r in range(a.rounds):
    for v in variants:
        for name, src, isb in (("fp32", h, False), ("bf16", hb, True)):
            tot = 0.0
            for _ in range(a.reps):
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(*v, src, isb)
                e1.record()
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            times[(v, name)].append(tot / a.reps * 1e3)
hip.set_option("lnq_ring", 0)
hip.set_option("fr_rot", 1)
print(f"gemm_lnq, M = {M}, random data, h flushed from the Infinity Cache before every launch; us per launch (median of "
      f"{a.rounds} rounds x {a.reps} launches, min in brackets); 38.65 GFLOP per launch")
for (v, name), ts in times.items():
    ts = sorted(ts)
    med = ts[len(ts) // 2]
    print(f"  MFMA {'32x32x16' if v[0] == 32 else '16x16x32'}  ring {v[1]} stages  h {name}:  {med:7.1f} us  [{ts[0]:7.1f}]  "
          f"{2.0 * M * d * d / (med * 1e-6) / 1e12:6.1f} TFLOP/s")

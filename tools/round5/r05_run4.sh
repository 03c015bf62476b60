#!/bin/bash
# round-5 GPU batch 4: attn64v2 with its eight K fragments requested together (KPF, attn_flags 65536): bits, microbench, in-model A/B
mkdir -p gpurun_out
{
python - <<'PY'
import math, torch, ctypes as C
from ditto_tts_amd import hip
lib = hip.lib(); st = torch.cuda.current_stream().cuda_stream
B, H, Sq, Skv, dh = 2, 12, 1024, 1024, 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(3)
q = (torch.randn(B * Sq, d, device="cuda", generator=g) * (1.4426950408889634 / 8)).to(torch.bfloat16)
k = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
outs = []
for fl in (16 + 256, 16 + 256 + 65536):
    hip.set_option("attn_flags", fl)
    o = torch.empty_like(q)
    hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, o.data_ptr(), d, B, H, Sq, Skv, dh, 0.125, None, 0, st))
    outs.append(o)
hip.set_option("attn_flags", 3)
print("KPF bitwise equal to attn64v2:", bool(torch.equal(outs[0], outs[1])), "finite:", bool(torch.isfinite(outs[1].float()).all()))
PY
echo "== microbench, C2 shape (16 = attn64v2, 65552 = + K fragments requested together)"
timeout 300 python tools/attn_bench.py --variants 16,65552 --rounds 7 --iters 10 2>&1 | tail -n 2
echo "== in the model, C2 B = 32 (@attn_flags)"
timeout 600 python tools/step_ab.py --variants "0/321@0~3,0/321@65536~3" --rounds 5 --steps 5 2>&1 | tail -n 3
} > gpurun_out/r05_kpf_ab.txt 2>&1
cut -c1-200 gpurun_out/r05_kpf_ab.txt

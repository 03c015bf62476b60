#!/bin/bash
# round-5 GPU batch 6: attn64w4 with 256 queries per workgroup (experimental build); stamp timeline of the fused norm2 + q-projection
mkdir -p gpurun_out
export DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_exp.so
{
python - <<'PY'
import math, torch
from ditto_tts_amd import hip
lib = hip.lib(); st = torch.cuda.current_stream().cuda_stream
for (B, H, Sq, Skv) in ((2, 12, 1024, 1024), (1, 3, 300, 200), (1, 2, 333, 2048)):
    dh = 64; d = H * dh
    g = torch.Generator(device="cuda").manual_seed(3)
    q = (torch.randn(B * Sq, d, device="cuda", generator=g) * (1.4426950408889634 / 8)).to(torch.bfloat16)
    k = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B * Skv, d, device="cuda", generator=g).to(torch.bfloat16)
    outs = []
    for fl in (16 + 256, 16 + 32768, 16 + 32768 + 131072):
        hip.set_option("attn_flags", fl)
        o = torch.empty_like(q)
        hip.check(lib.ditto_attention_bf16(q.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, o.data_ptr(), d, B, H, Sq, Skv, dh, 0.125, None, 0, st))
        outs.append(o.float())
    hip.set_option("attn_flags", 3)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print((B, H, Sq, Skv), "w4 vs v2 rel-L2", rel(outs[1], outs[0]), " w4 with 256 queries per workgroup vs w4: bitwise", bool(torch.equal(outs[2], outs[1])))
PY
echo "== microbench, C2 shape (16 = attn64v2, 49168 = w4, 180240 = w4 with 256 queries per workgroup)"
timeout 300 python tools/attn_bench.py --variants 16,49168,180240 --rounds 7 --iters 10 2>&1 | tail -n 3
echo "== C4 shape (Sq = Skv = 4096, B = 8)"
timeout 300 python tools/attn_bench.py --variants 16,49168,180240 --B 8 --Sq 4096 --Skv 4096 --rounds 5 --iters 5 2>&1 | tail -n 3
echo "== in the model, C2 B = 32 (@attn_flags: 0 = shipped, 16384 = w4, 147456 = w4 with 256 queries per workgroup)"
timeout 600 python tools/step_ab.py --variants "0/321@0~3,0/321@16384~3,0/321@147456~3" --rounds 5 --steps 5 2>&1 | tail -n 4
} > gpurun_out/r05_w4x8_ab.txt 2>&1
cut -c1-200 gpurun_out/r05_w4x8_ab.txt
unset DITTO_HIP_LIB
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_lnqstamp.so timeout 300 python tools/lnq_stamps.py > gpurun_out/r05_lnq_stamps.txt 2>&1
tail -n 8 gpurun_out/r05_lnq_stamps.txt

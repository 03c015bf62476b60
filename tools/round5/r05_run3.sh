#!/bin/bash
# round-5 GPU batch 3: attn64w4 (four waves per SIMD): unit parity, microbench against attn64v2, in-model A/B; attn64v2 stamp timeline
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "test_attention" 2>&1 | tail -n 4
{
echo "== microbench, C2 cross / self shape (attn_flags 16 = attn64v2 at 3 waves per SIMD, 16400 = attn64w4 on large grids)"
timeout 300 python tools/attn_bench.py --variants 16,16400 --rounds 7 --iters 10 2>&1 | tail -n 2
echo "== C4 shape (Sq = Skv = 4096, B = 8)"
timeout 300 python tools/attn_bench.py --variants 16,16400 --B 8 --Sq 4096 --Skv 4096 --rounds 5 --iters 5 2>&1 | tail -n 2
echo "== in the model, C2 B = 32 (@attn_flags)"
timeout 600 python tools/step_ab.py --variants "0/321@0~3,0/321@16384~3" --rounds 5 --steps 5 2>&1 | tail -n 3
echo "== in the model, C2 B = 16"
timeout 600 python tools/step_ab.py --batch 16 --variants "0/321@0~3,0/321@16384~3" --rounds 4 --steps 5 2>&1 | tail -n 3
} > gpurun_out/r05_w4_ab.txt 2>&1
cut -c1-200 gpurun_out/r05_w4_ab.txt
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_a2stamp.so timeout 300 python tools/a2_stamps.py > gpurun_out/r05_a2_stamps.txt 2>&1
cat gpurun_out/r05_a2_stamps.txt | tail -n 12

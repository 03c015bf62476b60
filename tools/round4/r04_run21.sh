#!/bin/bash
# round-4 GPU batch 21: gemm256 tile timeline (s_memtime stamps, diagnostic build), per wave group, flat K loop (default) / round-3 tile switch
mkdir -p gpurun_out
rm -f gpurun_out/r04_g256_stamps_flat.txt
export DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_g256stamp.so
for fl in 321 16705; do
  timeout 300 python tools/g256_stamps.py --flags $fl --shapes gated,qkv >> gpurun_out/r04_g256_stamps_flat.txt 2>&1
done
cat gpurun_out/r04_g256_stamps_flat.txt

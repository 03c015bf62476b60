#!/bin/bash
# round-5 GPU batch 9: stamp timeline of the 128-row full-row GEMM + residual + LayerNorm kernel (out-projection + norm3, fc2 + norm1)
mkdir -p gpurun_out
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_frdstamp.so timeout 300 python tools/frd_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_frd_stamps.txt
cat gpurun_out/r05_frd_stamps.txt

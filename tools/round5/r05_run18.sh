#!/bin/bash
mkdir -p gpurun_out
{
echo "== C5_bf16: tile rule vs 256 x 256 forced (one round of 256 tiles for the N = 1024 GEMMs at M = 16384), fr_mask 3 / 1 / 0"
timeout 600 python tools/step_ab.py --config C5_bf16 --variants "0/321~3,256/321~3,256/321~1,256/321" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 5
} > gpurun_out/r05_c5_tile256.txt 2>&1
cat gpurun_out/r05_c5_tile256.txt

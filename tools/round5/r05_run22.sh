#!/bin/bash
mkdir -p gpurun_out
{
echo "== C5 (fp8 linears): fr_mask 3 / 0, tile 256 forced"
timeout 600 python tools/step_ab.py --config C5 --variants "0/321~3,0/321,256/321~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 4
} > gpurun_out/r05_c5_fp8_ab.txt 2>&1
cat gpurun_out/r05_c5_fp8_ab.txt

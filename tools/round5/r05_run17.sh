#!/bin/bash
mkdir -p gpurun_out
{
echo "== C5_bf16: fr_mask 3 / 1 / 0 / 2"
timeout 600 python tools/step_ab.py --config C5_bf16 --variants "0/321~3,0/321~1,0/321,0/321~2" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 5
} > gpurun_out/r05_c5_frmask.txt 2>&1
cat gpurun_out/r05_c5_frmask.txt

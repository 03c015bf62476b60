#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "qkv_gemm_split" 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
for b in 8 16 24 32; do
echo "== C2 shapes, B = $b: qkv_split 0 / 64"
timeout 300 python tools/step_ab.py --batch $b --variants "0/321~3:qkv_split.0,0/321~3:qkv_split.64" --rounds 4 2>&1 | grep -v amdgpu.ids | tail -n 3
done
} > gpurun_out/r05_qkv_split_ab.txt 2>&1
cat gpurun_out/r05_qkv_split_ab.txt

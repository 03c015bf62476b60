#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
echo "== B = 1, 2: ll_mask 3 (round 4) / 7 (+ attention split over the keys)"
for b in 1 2; do timeout 300 python tools/step_ab.py --batch $b --variants "0/321~3:ll_mask.3,0/321~3:ll_mask.7" --rounds 4 2>&1 | grep -v amdgpu.ids | tail -n 3; done
} > gpurun_out/r05_splitkv.txt 2>&1
cat gpurun_out/r05_splitkv.txt

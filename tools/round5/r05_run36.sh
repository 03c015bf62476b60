#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "q_projection or lnq or layernorm_fused" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
echo "== C2 B = 32: lnq_persist 0 / 1 in one process"
timeout 600 python tools/step_ab.py --variants "0/321~3:lnq_persist.0,0/321~3:lnq_persist.1" --rounds 4 2>&1 | grep -v amdgpu.ids | tail -n 3
echo "== C4 (N = 4096, B = 8)"
timeout 600 python tools/step_ab.py --config C4 --variants "0/321~3:lnq_persist.0,0/321~3:lnq_persist.1" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 3
} > gpurun_out/r05_lnq_persist.txt 2>&1
cat gpurun_out/r05_lnq_persist.txt

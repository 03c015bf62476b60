#!/bin/bash
# round-5 GPU batch 10: the fused norm2 + q-projection on TWO waves per SIMD (lnq_waves 8): bits, in-model A/B in one process
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "two_waves_per_simd" 2>&1 | tail -n 3
timeout 900 python tools/step_ab.py --variants "0/321~3:lnq_waves.4,0/321~3:lnq_waves.8" --rounds 6 --steps 5 > gpurun_out/r05_step_ab_lnq8.txt 2>&1
tail -n 3 gpurun_out/r05_step_ab_lnq8.txt | cut -c1-200

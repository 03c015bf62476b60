#!/bin/bash
# round-4 GPU batch 23: issue priority swapped between the two waves of a SIMD halfway through gemm256's fast epilogue (gemm_flags 32768)
mkdir -p gpurun_out
timeout 900 python tools/step_ab.py --variants "0/321~3,0/33089~3" --rounds 6 --steps 10 > gpurun_out/r04_step_ab_epi_prio.txt 2>&1
tail -n 3 gpurun_out/r04_step_ab_epi_prio.txt | cut -c1-150
timeout 900 python tools/step_ab.py --variants "0/33089~3,0/321~3" --rounds 6 --steps 10 >> gpurun_out/r04_step_ab_epi_prio.txt 2>&1
tail -n 2 gpurun_out/r04_step_ab_epi_prio.txt | cut -c1-150

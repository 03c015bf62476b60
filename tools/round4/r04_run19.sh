#!/bin/bash
# round-4 GPU batch 19: flat K loop v2, longer A/B of the denoise step (8 rounds x 10 steps, two orders), B = 8 and C5
mkdir -p gpurun_out
timeout 900 python tools/step_ab.py --variants "0/321~3,0/16705~3" --rounds 8 --steps 10 > gpurun_out/r04_step_ab_flat2_long.txt 2>&1
tail -n 2 gpurun_out/r04_step_ab_flat2_long.txt | cut -c1-150
timeout 900 python tools/step_ab.py --variants "0/16705~3,0/321~3" --rounds 8 --steps 10 >> gpurun_out/r04_step_ab_flat2_long.txt 2>&1
tail -n 2 gpurun_out/r04_step_ab_flat2_long.txt | cut -c1-150
timeout 600 python tools/step_ab.py --variants "0/321~3,0/16705~3" --rounds 6 --steps 10 --batch 8 > gpurun_out/r04_step_ab_flat2_b8.txt 2>&1
tail -n 2 gpurun_out/r04_step_ab_flat2_b8.txt | cut -c1-150
timeout 600 python tools/step_ab.py --variants "0/321~3,0/16705~3" --rounds 6 --steps 10 --batch 16 > gpurun_out/r04_step_ab_flat2_b16.txt 2>&1
tail -n 2 gpurun_out/r04_step_ab_flat2_b16.txt | cut -c1-150

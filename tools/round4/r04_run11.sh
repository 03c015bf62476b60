#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_t11.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t11.log
grep -E "^FAILED|passed|failed|rc=" gpurun_out/r04_t11.log | tail -n 12
timeout 600 python tools/step_ab.py --batch 1 --variants "0/321~3,0/321#-1~3,0/321#256~3" --rounds 4 --steps 10 > gpurun_out/r04_ab_b1_v2.txt 2>&1
tail -n 4 gpurun_out/r04_ab_b1_v2.txt | cut -c1-220
timeout 600 python tools/step_ab.py --batch 2 --variants "0/321~3,0/321#-1~3,0/321#256~3" --rounds 4 --steps 10 > gpurun_out/r04_ab_b2_v2.txt 2>&1
tail -n 4 gpurun_out/r04_ab_b2_v2.txt | cut -c1-220

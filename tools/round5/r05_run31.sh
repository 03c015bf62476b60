#!/bin/bash
mkdir -p gpurun_out
{
echo "== C2 shapes, B = 1 and B = 2 (low-latency class): per-class us"
for b in 1 2; do timeout 300 python tools/step_ab.py --batch $b --variants "0/321~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 2; done
} > gpurun_out/r05_b1_classes.txt 2>&1
cat gpurun_out/r05_b1_classes.txt

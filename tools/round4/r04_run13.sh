#!/bin/bash
mkdir -p gpurun_out
for b in 4 8 12 16; do
  echo "== B = $b"; timeout 600 python tools/step_ab.py --batch $b --variants "0/321~3,0/321~3:lnq_min_rows.2049" --rounds 4 --steps 8 2>&1 | tail -n 3 | cut -c1-230
done > gpurun_out/r04_ab_lnq_mid.txt 2>&1
cat gpurun_out/r04_ab_lnq_mid.txt

#!/bin/bash
mkdir -p gpurun_out
{
for b in 4 8 12 16; do
echo "== C2 shapes, B = $b: norm2 + q-projection fused below the full-row class (lnq_min_rows 2049), 8 and 4 waves"
timeout 300 python tools/step_ab.py --batch $b --variants "0/321~3,0/321~3:lnq_min_rows.2049,0/321~3:lnq_min_rows.2049;lnq_waves.4" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 4
done
for c in C5_bf16 C5; do
echo "== $c"
timeout 300 python tools/step_ab.py --config $c --variants "0/321~3,0/321~3:lnq_waves.4" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 3
done
} > gpurun_out/r05_small_batch_lnq8.txt 2>&1
cat gpurun_out/r05_small_batch_lnq8.txt

#!/bin/bash
# round-4 GPU batch 18: the flat K loop of gemm256, second version (explicit source bases, one switch block): bitwise test,
# then the denoise step with gemm_flags 321 / 16705 in one process (step_ab), then the training step both ways.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "flat_k or gemm" > gpurun_out/r04_t18.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t18.log
tail -n 4 gpurun_out/r04_t18.log
timeout 900 python tools/step_ab.py --variants "0/321~3,0/16705~3" --rounds 4 --steps 5 > gpurun_out/r04_step_ab_flat2.txt 2>&1
tail -n 3 gpurun_out/r04_step_ab_flat2.txt
for i in 1 2; do
  for fl in 321 16705; do
    timeout 300 python tools/train_report.py --batch 32 --steps 4 --gemm-flags $fl 2>&1 | tail -n 1 | sed "s/^/gemm_flags $fl: /" >> gpurun_out/r04_train_flat2_ab.txt
  done
done
cut -c1-160 gpurun_out/r04_train_flat2_ab.txt

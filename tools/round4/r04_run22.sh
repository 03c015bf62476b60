#!/bin/bash
# round-4 GPU batch 22: flat K loop with the wave-group stagger persisting over the tile switch: bitwise tests, stamps, and the
# denoise / training step against the previous commit's library (alternating processes on one box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "flat_k or gemm" > gpurun_out/r04_t22.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t22.log
tail -n 3 gpurun_out/r04_t22.log
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_g256stamp.so timeout 300 python tools/g256_stamps.py --flags 321 --shapes gated,qkv > gpurun_out/r04_g256_stamps_flat_skew.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_g256_stamps_flat_skew.txt
rm -f gpurun_out/r04_step_ab_flat_skew.txt
for i in 1 2 3; do
  DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_prev.so timeout 300 python tools/step_ab.py --variants "0/321~3" --rounds 6 --steps 10 2>&1 | tail -n 1 | cut -c1-150 | sed "s/^/prev /" >> gpurun_out/r04_step_ab_flat_skew.txt
  timeout 300 python tools/step_ab.py --variants "0/321~3" --rounds 6 --steps 10 2>&1 | tail -n 1 | cut -c1-150 | sed "s/^/new  /" >> gpurun_out/r04_step_ab_flat_skew.txt
done
cat gpurun_out/r04_step_ab_flat_skew.txt
for i in 1 2; do
  DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_prev.so timeout 300 python tools/train_report.py --batch 32 --steps 4 2>&1 | tail -n 1 | cut -c1-150 | sed "s/^/prev /" >> gpurun_out/r04_train_flat_skew.txt
  timeout 300 python tools/train_report.py --batch 32 --steps 4 2>&1 | tail -n 1 | cut -c1-150 | sed "s/^/new  /" >> gpurun_out/r04_train_flat_skew.txt
done
cat gpurun_out/r04_train_flat_skew.txt

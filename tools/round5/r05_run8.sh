#!/bin/bash
# round-5 GPU batch 8: fused norm2 + q-projection with all 16 rows of a wave requested at once: parity, stamps, in-model against r04
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "lnq or q_projection or norm2 or g2 or g8" 2>&1 | tail -n 3
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_lnqstamp.so timeout 300 python tools/lnq_stamps.py 2>&1 | tail -n 5 > gpurun_out/r05_lnq_stamps_v3.txt
cat gpurun_out/r05_lnq_stamps_v3.txt
{
for i in 1 2; do
echo "== frozen round-4 library"; DITTO_HIP_LIB=$PWD/build/libditto_r04.so timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
echo "== this build"; timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
done
} > gpurun_out/r05_step_ab_lnq16.txt 2>&1
cut -c1-200 gpurun_out/r05_step_ab_lnq16.txt

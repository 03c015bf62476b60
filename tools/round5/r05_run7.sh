#!/bin/bash
# round-5 GPU batch 7: wave reductions on DPP instead of ds_bpermute (csrc/common.h wave_sum / wave_max): every test that pins a
# LayerNorm, the fused q-projection's stamp timeline again, in-model step against the frozen round-4 library
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_slp.py -q -x 2>&1 | tail -n 4
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_lnqstamp.so timeout 300 python tools/lnq_stamps.py 2>&1 | tail -n 5 > gpurun_out/r05_lnq_stamps_dpp.txt
cat gpurun_out/r05_lnq_stamps_dpp.txt
{
for i in 1 2; do
echo "== frozen round-4 library"; DITTO_HIP_LIB=$PWD/build/libditto_r04.so timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
echo "== this build"; timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
done
} > gpurun_out/r05_step_ab_dpp.txt 2>&1
cut -c1-200 gpurun_out/r05_step_ab_dpp.txt

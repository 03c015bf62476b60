#!/bin/bash
mkdir -p gpurun_out
for k in 768 3072; do
echo "== K = $k"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_frdstamp$k.so timeout 300 python tools/frd_stamps_model.py 2>&1 | grep -v amdgpu.ids | tail -n 7
done > gpurun_out/r05_frd_stamps_model.txt
cat gpurun_out/r05_frd_stamps_model.txt

#!/bin/bash
mkdir -p gpurun_out
for lib in libditto_diag_frdstamp768.so libditto_diag_frdnores768.so; do
echo "== $lib"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/$lib timeout 300 python tools/frd_stamps_model.py 2>&1 | grep -v amdgpu.ids | tail -n 6
done > gpurun_out/r05_frd_nores_stamps.txt
cat gpurun_out/r05_frd_nores_stamps.txt

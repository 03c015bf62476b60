#!/bin/bash
mkdir -p gpurun_out
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_lnqstamp.so timeout 300 python tools/frq_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_frq_stamps.txt
cat gpurun_out/r05_frq_stamps.txt

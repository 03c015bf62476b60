#!/bin/bash
mkdir -p gpurun_out
for w in 8 4; do
echo "== lnq_waves $w"; DITTO_LNQ_WAVES=$w DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_lnqstamp.so timeout 300 python tools/lnq_stamps.py --hot 2>&1 | grep -v amdgpu.ids | tail -n 5
done > gpurun_out/r05_lnq_stamps_fine.txt
cat gpurun_out/r05_lnq_stamps_fine.txt

#!/bin/bash
mkdir -p gpurun_out
DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_qkvplain.so timeout 600 python tools/step_ab.py --variants "0/321~3,192/321~3,256/321~3" --rounds 4 --steps 5 > gpurun_out/r04_ab_qkvplain.txt 2>&1
tail -n 4 gpurun_out/r04_ab_qkvplain.txt | cut -c1-220

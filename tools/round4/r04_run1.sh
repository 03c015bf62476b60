#!/bin/bash
# round-4 GPU batch 1: the whole -m gpu suite, then same-box A/Bs of the denoise step (fp32 vs bf16 residual stream, and the
# frozen round-3 library), then the training step against the frozen library.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t1.log
tail -4 gpurun_out/r04_t1.log
timeout 600 python tools/step_ab.py --variants "0/321~3,0/321~3:residual_bf16.1" --rounds 4 --steps 5 > gpurun_out/r04_ab_hb.txt 2>&1
tail -3 gpurun_out/r04_ab_hb.txt
DITTO_HIP_LIB=$PWD/build/libditto_r03.so timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 4 --steps 5 > gpurun_out/r04_ab_r03.txt 2>&1
tail -2 gpurun_out/r04_ab_r03.txt
timeout 600 python tools/train_report.py --batch 32 --steps 3 > gpurun_out/r04_train1.log 2>&1
DITTO_HIP_LIB=$PWD/build/libditto_r03.so timeout 600 python tools/train_report.py --batch 32 --steps 3 > gpurun_out/r04_train1_r03.log 2>&1
tail -3 gpurun_out/r04_train1.log gpurun_out/r04_train1_r03.log

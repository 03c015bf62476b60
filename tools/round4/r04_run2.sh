#!/bin/bash
# round-4 GPU batch 2: the new kernels' tests first, then the rest of the suite, then same-box A/Bs of the denoise step.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused" > gpurun_out/r04_t2a.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t2a.log
tail -n 4 gpurun_out/r04_t2a.log
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "bf16_residual or norm2_fused" > gpurun_out/r04_t2b.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t2b.log
grep -E "rel-L2|passed|failed|rc=" gpurun_out/r04_t2b.log | tail -n 14
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -m gpu -q > gpurun_out/r04_t2c.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t2c.log
tail -n 4 gpurun_out/r04_t2c.log
timeout 900 python tools/step_ab.py --variants "0/321~3,0/321~3:residual_bf16.1,0/321~3:lnq.32,0/321~3:lnq.16,0/321~3:residual_bf16.1;lnq.32,0/321~3:residual_bf16.1;lnq.16" --rounds 4 --steps 5 > gpurun_out/r04_ab_lnq.txt 2>&1
tail -n 8 gpurun_out/r04_ab_lnq.txt

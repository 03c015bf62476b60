#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "head_dims_that or padded_heads" > gpurun_out/r04_t10.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t10.log
grep -E "rel-L2|passed|failed|rc=|^E  " gpurun_out/r04_t10.log | tail -n 20 | cut -c1-250

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "fused or c5" > gpurun_out/r04_t15.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t15.log
grep -E "^FAILED|passed|failed|rc=|^E  " gpurun_out/r04_t15.log | tail -n 10 | cut -c1-250
for cfg in C5 C5_bf16; do
  for lnq in 0 32; do
    echo "== $cfg lnq=$lnq"; DITTO_LNQ=$lnq timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-sweep --no-c3 --no-other-configs --no-parity --loops 2 --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_classes'].items()})"
  done
done > gpurun_out/r04_c5_lnq.txt 2>&1
cat gpurun_out/r04_c5_lnq.txt | cut -c1-400

#!/bin/bash
# round-4 GPU batch 25: the training step on the bf16 residual stream: training tests, step A/B (train_flags 16 = fp32 tape)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r04_t25.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t25.log
tail -n 12 gpurun_out/r04_t25.log
rm -f gpurun_out/r04_train_stream_bf16_ab.txt
for i in 1 2 3; do
  for f in 16 0; do
    timeout 300 python tools/train_report.py --batch 32 --steps 4 --train-flags $f 2>&1 | tail -n 1 | cut -c1-150 | sed "s/^/flags $f: /" >> gpurun_out/r04_train_stream_bf16_ab.txt
  done
done
cat gpurun_out/r04_train_stream_bf16_ab.txt
python -c "
from oracle.checks import train_grad_parity
import json
print(json.dumps({k: v for k, v in train_grad_parity('cuda').items() if k != 'what'}))" 2>&1 | tail -n 1

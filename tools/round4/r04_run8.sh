#!/bin/bash
# round-4 GPU batch 8: training step after the gated-backward math change: gradient tests, step time against the frozen round-3
# library (alternating processes), kernel stats
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q > gpurun_out/r04_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t8.log
tail -n 3 gpurun_out/r04_t8.log
for i in 1 2; do
  timeout 600 python tools/train_report.py --batch 32 --steps 4 2>&1 | tail -n 1 | cut -c1-220
  DITTO_HIP_LIB=$PWD/build/libditto_r03.so timeout 600 python tools/train_report.py --batch 32 --steps 4 2>&1 | tail -n 1 | cut -c1-220
done > gpurun_out/r04_train_ab.txt 2>&1
cat gpurun_out/r04_train_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_train -o kt -- python3 tools/train_report.py --batch 32 --steps 3 > gpurun_out/kt_train.log 2>&1
cp $(find gpurun_out/kt_train -name "*kernel_stats.csv" | head -1) gpurun_out/r04_train_b32_kernel_stats.csv
rm -rf gpurun_out/kt_train
head -n 30 gpurun_out/r04_train_b32_kernel_stats.csv | cut -c1-200

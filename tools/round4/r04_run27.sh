#!/bin/bash
# round-4 GPU batch 27: bf16 column sums with 16-B loads, four rows in flight: training tests, kernel table of the step
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r04_t27.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t27.log
tail -n 3 gpurun_out/r04_t27.log
for i in 1 2; do timeout 300 python tools/train_report.py --batch 32 --steps 4 2>&1 | tail -n 1 | cut -c1-150; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_train27_prof -o t27 -- python3 $GRAFT_REPO_ROOT/tools/train_report.py --batch 32 --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/r04_train27_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_db.py gpurun_out/r04_train27_prof/t27_results.db 45 | cut -c1-150 > gpurun_out/r04_train_b32_kernel_stats_v2.txt
cat gpurun_out/r04_train_b32_kernel_stats_v2.txt

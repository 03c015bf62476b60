#!/bin/bash
# round-4 GPU batch 16: the gated-MLP backward as the fc2 dgrad's epilogue (EPI_GATED_BWD) + the training forward's gated GEMM
# as its own instantiation (EPI_GATED_PRE): the training tests, then the training step with train_flags 6 (old) / 0 (new),
# alternating processes on one box, then a kernel trace of the new step.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r04_t16.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t16.log
tail -n 6 gpurun_out/r04_t16.log
for i in 1 2; do
  for f in 6 0 2 4; do
    timeout 300 python tools/train_report.py --batch 32 --steps 4 --train-flags $f 2>&1 | tail -n 1 | sed "s/^/flags $f: /" >> gpurun_out/r04_train_gated_ab.txt
  done
done
cat gpurun_out/r04_train_gated_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_train16_prof -o t16 -- python3 $GRAFT_REPO_ROOT/tools/train_report.py --batch 32 --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/r04_train16_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py gpurun_out/r04_train16_prof 2>/dev/null | head -n 40 > gpurun_out/r04_train16_kernel_stats.txt
head -n 30 gpurun_out/r04_train16_kernel_stats.txt

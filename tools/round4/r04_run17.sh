#!/bin/bash
# round-4 GPU batch 17: EPI_GATED_BWD with 24 loads in flight per wave; start stagger of the two training-epilogue GEMMs
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "gated_mlp_backward or timed_dimensions or bit_reproducible" > gpurun_out/r04_t17.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t17.log
tail -n 4 gpurun_out/r04_t17.log
rm -f gpurun_out/r04_train_stagger_ab.txt
for i in 1 2; do
  for st in 0 400 800 1300 2000; do
    DITTO_TRAIN_STAGGER=$st timeout 300 python tools/train_report.py --batch 32 --steps 4 2>&1 | tail -n 1 | sed "s/^/stagger $st: /" >> gpurun_out/r04_train_stagger_ab.txt
  done
  timeout 300 python tools/train_report.py --batch 32 --steps 4 --train-flags 6 2>&1 | tail -n 1 | sed "s/^/flags 6: /" >> gpurun_out/r04_train_stagger_ab.txt
done
cut -c1-150 gpurun_out/r04_train_stagger_ab.txt
cd /tmp && export TMPDIR=/tmp
DITTO_TRAIN_STAGGER=800 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_train17_prof -o t17 -- python3 $GRAFT_REPO_ROOT/tools/train_report.py --batch 32 --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/r04_train17_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_db.py gpurun_out/r04_train17_prof/t17_results.db 12 | cut -c1-130

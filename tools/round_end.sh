#!/bin/bash
# What a round ends with, in one gpurun call (the CPU suite runs HERE too, next to the GPU suite — round 5 shipped
# three "final tree validated" commits on a tree whose CPU suite was red):
#   gpurun --timeout 3000 -- 'tools/round_end.sh r06_final'
# 1. python -m pytest -m "not gpu"   2. python -m pytest -m gpu   3. every rocprofv3 pass of tools/profile_round.sh (which starts with
# the full bench.py run).  Logs under gpurun_out/round_end_<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-final}
OUT=gpurun_out/round_end_$TAG
mkdir -p $OUT
python -m pytest tests -q -m "not gpu" > $OUT/cpu_suite.log 2>&1; echo "cpu suite rc=$?" | tee $OUT/rc.txt; tail -2 $OUT/cpu_suite.log
python -m pytest tests -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/rc.txt; tail -2 $OUT/gpu_suite.log
bash tools/profile_round.sh $TAG

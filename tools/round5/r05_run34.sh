#!/bin/bash
mkdir -p gpurun_out/prof_split
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split/kt -o kt -- python3 tools/attn_split_bench.py > gpurun_out/r05_attn_split_bench.txt 2>&1
grep "ll_mask" gpurun_out/r05_attn_split_bench.txt
cat $(find gpurun_out/prof_split/kt -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -8
rm -rf gpurun_out/prof_split

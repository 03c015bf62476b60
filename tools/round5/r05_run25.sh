#!/bin/bash
# kernel-trace + stats pass once more WITHOUT the parity leg (whose launches follow seconds of CPU work: cold clocks, max 393 us for the gated GEMM)
mkdir -p gpurun_out/prof_r05_v1b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r05_v1b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline --profile-steps 5 --loops 0 --no-sweep --no-c3 --no-other-configs --no-parity --steps 10 --warmup 2 > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r05_v1b_kernel_stats.csv
tail -1 $OUT/kt.log > gpurun_out/r05_v1b_bench_under_rocprof.json
rm -rf $OUT/kt
grep -E "gemm256_kernel<3" gpurun_out/r05_v1b_kernel_stats.csv | cut -c1-200
python -c "
import json; b=json.load(open('gpurun_out/r05_v1b_bench_under_rocprof.json')); print(b['ms_per_step'], b['roofline']['avg_launch_ms'])"

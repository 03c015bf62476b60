#!/bin/bash
# kernel-trace + stats summaries of the round's last tree for the configurations beside the headline: C5 (fp8 / bf16), B = 1, B = 8, training step
mkdir -p gpurun_out/kt5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BF="--no-cpu-baseline --profile-steps 0 --loops 0 --no-sweep --no-c3 --no-other-configs --no-parity"
run() {  # name, command...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt5/$name -o kt -- "$@" > gpurun_out/kt5/$name.log 2>&1
  cp $(find gpurun_out/kt5/$name -name "*kernel_stats.csv" | head -1) gpurun_out/r05_${name}_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/kt5/$name
  echo "== $name"; head -9 gpurun_out/r05_${name}_kernel_stats.csv | cut -c1-170
}
run c5 python3 bench.py --config C5 $BF --steps 10 --warmup 2
run c5_bf16 python3 bench.py --config C5_bf16 $BF --steps 10 --warmup 2
run b1 python3 bench.py --batch 1 $BF --steps 20 --warmup 5
run b8 python3 bench.py --batch 8 $BF --steps 20 --warmup 5
run train_b32 python3 tools/train_report.py --batch 32 --steps 3

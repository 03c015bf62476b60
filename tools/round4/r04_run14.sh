#!/bin/bash
# round-4 GPU batch 14: the whole suite on the final build, the complete profile set (r04_v3), smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_t14.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t14.log
grep -E "^FAILED|passed|failed|rc=" gpurun_out/r04_t14.log | tail -n 8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash tools/profile_round.sh r04_v3 > gpurun_out/r04_profile_round3.log 2>&1
tail -n 14 gpurun_out/r04_profile_round3.log | cut -c1-200

#!/bin/bash
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5
echo "== C5_bf16 / C5 (bench.py --config), new rules"
for c in C5_bf16 C5; do timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-c3 --no-parity --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config'], d['ms_per_step'])"; done
echo "== sweep"
timeout 600 python tools/step_ab.py --batch 8 --variants "0/321~3,0/321~3:lnq_min_rows.0" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 3
} > gpurun_out/r05_t19.txt 2>&1
cat gpurun_out/r05_t19.txt

#!/bin/bash
# round-5 GPU batch 11: lnq_waves 8 as the default (d = 768 and d = 1024): the whole -m gpu suite; C5 A/B; stamp timeline
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 4 > gpurun_out/r05_t11.txt; cat gpurun_out/r05_t11.txt
{
for w in 4 8; do
echo "== C5 fp8, DITTO_LNQ_WAVES=$w"; DITTO_LNQ_WAVES=$w timeout 600 python bench.py --config C5 --no-cpu-baseline --no-sweep --no-c3 --no-parity --steps 20 --loops 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernel_classes'].items()})"
done
for w in 4 8; do
echo "== C5 bf16, DITTO_LNQ_WAVES=$w"; DITTO_LNQ_WAVES=$w timeout 600 python bench.py --config C5_bf16 --no-cpu-baseline --no-sweep --no-c3 --no-parity --steps 20 --loops 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernel_classes'].items()})"
done
} > gpurun_out/r05_c5_lnq8_ab.txt 2>&1
cut -c1-260 gpurun_out/r05_c5_lnq8_ab.txt

#!/bin/bash
# round-4 GPU batch 6: K-loop rotation of the full-row kernels re-measured in the model (its W over-fetch shows in the PMC
# pass: fc2 reads 541 MB for 251 algorithmic), plus the bench line with the floor table
mkdir -p gpurun_out
timeout 900 python tools/step_ab.py --variants "0/321~3,0/321~3&0,0/321~3,0/321~3&0" --rounds 4 --steps 5 > gpurun_out/r04_ab_rot.txt 2>&1
tail -n 5 gpurun_out/r04_ab_rot.txt | cut -c1-230
python bench.py > gpurun_out/r04_bench_v2.json 2> gpurun_out/r04_bench_v2.err
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_v2.json'))
print(d['value'], d['ms_per_step'], d['step_frac_of_mfma_peak'], d.get('floor'), d['sweep'], [(o['config'], round(o['ms_per_step'],2)) for o in d['other_configs']])"

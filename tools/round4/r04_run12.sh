#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "low_latency or full_size_c2_properties or g2_ or g5_ or ragged" > gpurun_out/r04_t12.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t12.log
grep -E "^FAILED|passed|failed|rc=|^E  " gpurun_out/r04_t12.log | tail -n 12 | cut -c1-250
timeout 600 python tools/step_ab.py --batch 1 --variants "0/321~3:ll_mask.0,0/321~3:ll_mask.1,0/321~3:ll_mask.2,0/321~3:ll_mask.3" --rounds 4 --steps 10 > gpurun_out/r04_ab_b1_ll.txt 2>&1
tail -n 5 gpurun_out/r04_ab_b1_ll.txt | cut -c1-220
timeout 600 python tools/step_ab.py --batch 2 --variants "0/321~3:ll_mask.0,0/321~3:ll_mask.3" --rounds 4 --steps 10 > gpurun_out/r04_ab_b2_ll.txt 2>&1
tail -n 3 gpurun_out/r04_ab_b2_ll.txt | cut -c1-220

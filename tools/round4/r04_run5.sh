#!/bin/bash
# round-4 GPU batch 5: whole suite on the new defaults; the diagnostic builds of the floor table; B = 1 variants; the round's
# first complete profile set (tools/profile_round.sh r04_v1)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t5.log
grep -E "^FAILED|passed|failed|rc=" gpurun_out/r04_t5.log | tail -n 12
{
echo "== shipped"; timeout 600 python tools/step_ab.py --variants "0/321~3,0/325~3" --rounds 3 --steps 5 2>&1 | tail -n 3
for lib in hot nostore nodma; do
  echo "== libditto_diag_$lib.so"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_$lib.so timeout 600 python tools/step_ab.py --variants "0/321~3,0/325~3" --rounds 3 --steps 5 2>&1 | tail -n 3
done
} > gpurun_out/r04_floor_diag.txt 2>&1
cat gpurun_out/r04_floor_diag.txt | cut -c1-230
timeout 600 python tools/step_ab.py --batch 1 --variants "0/321~3,0/321#256~3,0/321^32!3~3,0/321#256^32!3~3" --rounds 4 --steps 10 > gpurun_out/r04_ab_b1.txt 2>&1
tail -n 5 gpurun_out/r04_ab_b1.txt | cut -c1-230
bash tools/profile_round.sh r04_v1 > gpurun_out/r04_profile_round.log 2>&1
tail -n 25 gpurun_out/r04_profile_round.log | cut -c1-200

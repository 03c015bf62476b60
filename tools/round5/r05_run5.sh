#!/bin/bash
# round-5 GPU batch 5: the two traffic knock-outs again, on VALID data (batch 2's builds multiplied uninitialised LDS / registers: NaN
# garbage downstream made every kernel of the step 10-16 % faster, and the knocked-out kernels' own gain could not be told from it)
mkdir -p gpurun_out
{
echo "== shipped"; timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
for lib in a2nodma lnqnow; do
  echo "== libditto_diag_$lib.so (valid data)"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_$lib.so timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
done
echo "== shipped (again)"; timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
} > gpurun_out/r05_floor_diag2.txt 2>&1
cut -c1-200 gpurun_out/r05_floor_diag2.txt

#!/bin/bash
mkdir -p gpurun_out
{
for lib in "" libditto_gated_h.so "" libditto_gated_h.so; do
  echo "== ${lib:-shipped (rational erf, packed fp32)}"
  if [ -n "$lib" ]; then export DITTO_HIP_LIB=$PWD/ditto_tts_amd/$lib; else unset DITTO_HIP_LIB; fi
  timeout 300 python tools/step_ab.py --variants "0/321~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 2
done
} > gpurun_out/r05_gated_h_ab.txt 2>&1
cat gpurun_out/r05_gated_h_ab.txt

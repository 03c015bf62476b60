#!/bin/bash
mkdir -p gpurun_out
{
for lib in build/libditto_r05.so "" build/libditto_r05.so ""; do
  echo "== ${lib:-this build (persistent fused q-projection)}"
  if [ -n "$lib" ]; then export DITTO_HIP_LIB=$PWD/$lib; else unset DITTO_HIP_LIB; fi
  timeout 300 python tools/step_ab.py --variants "0/321~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 2
done
} > gpurun_out/r05_lnq_persist_vs_frozen.txt 2>&1
cat gpurun_out/r05_lnq_persist_vs_frozen.txt

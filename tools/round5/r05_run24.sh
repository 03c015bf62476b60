#!/bin/bash
# gemm_frd accumulator-init window depth / group order, same box, alternating with the shipped library
mkdir -p gpurun_out
{
for lib in "" libditto_frd_w4p1.so libditto_frd_w8p0.so libditto_frd_w8p1.so libditto_frd_w12p1.so "" libditto_frd_w8p1.so libditto_frd_w12p1.so; do
  echo "== ${lib:-shipped}"
  if [ -n "$lib" ]; then export DITTO_HIP_LIB=$PWD/ditto_tts_amd/$lib; else unset DITTO_HIP_LIB; fi
  timeout 300 python tools/step_ab.py --variants "0/321~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 2
done
} > gpurun_out/r05_frd_window_ab.txt 2>&1
cat gpurun_out/r05_frd_window_ab.txt

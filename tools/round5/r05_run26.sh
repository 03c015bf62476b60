#!/bin/bash
mkdir -p gpurun_out
{
echo "== C2 B=32: tile rule vs 256 x 192 tiles forced where the epilogue exists (QKV: 1536 tiles = 6 whole rounds instead of 4.5 of 256 x 256)"
timeout 600 python tools/step_ab.py --variants "0/321~3,192/321~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 3
} > gpurun_out/r05_tile192_ab.txt 2>&1
cat gpurun_out/r05_tile192_ab.txt

#!/bin/bash
mkdir -p gpurun_out
{
echo "== C2 B=32, gemm_flags: 321 shipped, 323 = epilogue computes but does not store, 325 = no epilogue at all, 337 = stores to lane-linear addresses (diagnostic: wrong results, buffers keep valid numbers)"
timeout 600 python tools/step_ab.py --variants "0/321~3,0/323~3,0/325~3,0/337~3" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 5
} > gpurun_out/r05_epilogue_knockouts.txt 2>&1
cat gpurun_out/r05_epilogue_knockouts.txt

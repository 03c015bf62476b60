#!/bin/bash
mkdir -p gpurun_out
{
echo "== C2 B=32: RoPE angles computed in the QKV epilogue (321) vs cos / sin table loads (gemm_flags +1024 = 1345)"
timeout 600 python tools/step_ab.py --variants "0/321~3,0/1345~3" --rounds 4 2>&1 | grep -v amdgpu.ids | tail -n 3
} > gpurun_out/r05_rope_table_ab.txt 2>&1
cat gpurun_out/r05_rope_table_ab.txt

#!/bin/bash
mkdir -p gpurun_out
{
for b in 24 32; do
echo "== C2 shapes, B = $b: qkv_split 0 / 64 with the tail on the rule's kernel / on 256 x 128 tiles (129) / on 256 x 256 (256)"
timeout 400 python tools/step_ab.py --batch $b --variants "0/321~3:qkv_split.0,0/321~3:qkv_split.64,0/321~3:qkv_split.64;qkv_tail_tile.129,0/321~3:qkv_split.64;qkv_tail_tile.256" --rounds 3 2>&1 | grep -v amdgpu.ids | tail -n 5
done
} > gpurun_out/r05_qkv_split_tail_ab.txt 2>&1
cat gpurun_out/r05_qkv_split_tail_ab.txt

#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/pitch_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_pitch_ab.txt
cat gpurun_out/r05_pitch_ab.txt

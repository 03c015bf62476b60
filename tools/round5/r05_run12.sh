#!/bin/bash
# round-5 GPU batch 12: the cross out-projection + residual + norm3 on the fused q-projection kernel's skeleton ("frq"): unit parity,
# model parity (G2 / G8 / C2 headline shape against the oracle), in-model A/B in one process
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "out_projection_residual_layernorm" 2>&1 | grep -E "passed|failed|Error|assert" | tail -n 5
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "bf16_residual_stream or g8 or full_size_c2" 2>&1 | grep -E "passed|failed|rel-L2|Error|assert" | tail -n 12
timeout 900 python tools/step_ab.py --variants "0/321~3:frq.0,0/321~3:frq.1" --rounds 6 --steps 5 > gpurun_out/r05_step_ab_frq.txt 2>&1
tail -n 3 gpurun_out/r05_step_ab_frq.txt | cut -c1-200

#!/bin/bash
# round-4 GPU batch 4: defaults flipped (bf16 stream + fused norm2/q-proj): whole suite, A/Bs, the MFMA-shape probe
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_t4.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t4.log
grep -E "^FAILED|passed|failed|rc=" gpurun_out/r04_t4.log | tail -n 30
timeout 900 python tools/step_ab.py --variants "0/321~3:residual_bf16.0;lnq.0,0/321~3:residual_bf16.1;lnq.0,0/321~3:residual_bf16.1;lnq.32,0/321~3:residual_bf16.1;lnq.32;lnq_ring.4,0/321~3:residual_bf16.1;lnq.16,0/321~3:residual_bf16.1;lnq.16;lnq_ring.2,0/321%12~3:residual_bf16.1;lnq.32" --rounds 4 --steps 5 > gpurun_out/r04_ab_lnq2.txt 2>&1
tail -n 9 gpurun_out/r04_ab_lnq2.txt
timeout 600 python tools/step_ab.py --batch 8 --variants "0/321~3,0/321^10!3~3,0/321^10~3,0/321^2!3~3,0/321^8!3~3" --rounds 4 --steps 8 > gpurun_out/r04_ab_b8.txt 2>&1
tail -n 7 gpurun_out/r04_ab_b8.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_mfma_shape tools/probe_mfma_shape.hip 2>/dev/null
timeout 300 /tmp/probe_mfma_shape 40000 > gpurun_out/r04_mfma_shape_probe.txt 2>&1
cat gpurun_out/r04_mfma_shape_probe.txt

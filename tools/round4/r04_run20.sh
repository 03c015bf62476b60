#!/bin/bash
# round-4 GPU batch 20: flat K loop in the fp8 gemm256 instantiations: bitwise test, C5 fp8 / bf16 step with gemm_flags 321 / 16705
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fp8" > gpurun_out/r04_t20.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t20.log
tail -n 3 gpurun_out/r04_t20.log
for i in 1 2; do
for fl in 321 16705; do
  DITTO_GEMM_FLAGS=$fl timeout 300 python bench.py --config C5 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-c3 --no-parity --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 fp8 gemm_flags $fl:', round(d['ms_per_step'],3))" >> gpurun_out/r04_c5_flat_ab.txt
  DITTO_GEMM_FLAGS=$fl timeout 300 python bench.py --config C5_bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-c3 --no-parity --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 bf16 gemm_flags $fl:', round(d['ms_per_step'],3))" >> gpurun_out/r04_c5_flat_ab.txt
done
done
cat gpurun_out/r04_c5_flat_ab.txt

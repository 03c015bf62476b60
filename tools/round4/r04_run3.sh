#!/bin/bash
# round-4 GPU batch 3: gemm_lnq with the deeper ring / rotation; FETCH / WRITE calibration probe; B = 1 trace
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused" > gpurun_out/r04_t3a.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t3a.log
tail -n 3 gpurun_out/r04_t3a.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q -k "layernorm or norm2_fused or bf16_residual or g1_ or g2_ or adaln" > gpurun_out/r04_t3b.log 2>&1; echo "rc=$?" >> gpurun_out/r04_t3b.log
tail -n 3 gpurun_out/r04_t3b.log
timeout 900 python tools/step_ab.py --variants "0/321~3:residual_bf16.1,0/321~3:residual_bf16.1;lnq.32,0/321~3:residual_bf16.1;lnq.32;lnq_ring.4,0/321~3:residual_bf16.1;lnq.16,0/321~3:residual_bf16.1;lnq.16;lnq_ring.2,0/321~3%12:residual_bf16.1" --rounds 4 --steps 5 > gpurun_out/r04_ab_lnq2.txt 2>&1
tail -n 8 gpurun_out/r04_ab_lnq2.txt
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (counter passes carry no trace flags; the program follows `--`)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_fetch tools/probe_fetch_calib.hip
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_f -o f -- /tmp/probe_fetch > gpurun_out/calib_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_w -o w -- /tmp/probe_fetch > gpurun_out/calib_w.log 2>&1
python tools/fetch_calib_report.py $(find gpurun_out/calib_f -name "*counter_collection.csv" | head -1) $(find gpurun_out/calib_w -name "*counter_collection.csv" | head -1) > gpurun_out/r04_fetch_calib.txt 2>&1
cat gpurun_out/r04_fetch_calib.txt
rm -rf gpurun_out/calib_f gpurun_out/calib_w
# B = 1 and B = 8: kernel trace (what the step is made of at small batch)
for b in 1 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_b$b -o kt -- python3 bench.py --batch $b --no-cpu-baseline --profile-steps 0 --loops 0 --no-sweep --no-c3 --no-other-configs --no-parity --steps 20 --warmup 5 > gpurun_out/kt_b$b.log 2>&1
  cp $(find gpurun_out/kt_b$b -name "*kernel_stats.csv" | head -1) gpurun_out/r04_b${b}_kernel_stats.csv
  cp $(find gpurun_out/kt_b$b -name "*kernel_trace.csv" | head -1) gpurun_out/r04_b${b}_kernel_trace.csv
  python tools/gap_report.py gpurun_out/r04_b${b}_kernel_trace.csv > gpurun_out/r04_b${b}_gap_report.txt 2>&1
  tail -n 1 gpurun_out/kt_b$b.log | cut -c1-300
  head -n 12 gpurun_out/r04_b${b}_gap_report.txt
  rm -rf gpurun_out/kt_b$b
done
ls -la gpurun_out | tail -n 12

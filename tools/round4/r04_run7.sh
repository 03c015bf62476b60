#!/bin/bash
# round-4 GPU batch 7: cache-policy A/Bs of the operand streams (A nt / W nt / A sc1 in gemm256, A nt in gemm_frd) — alternating
# processes on one box, the shipped library first and last; then the fused q-projection kernel in isolation with wave-cycle counters
mkdir -p gpurun_out
{
for lib in hip diag_ant diag_wnt diag_asc1 diag_frdant hip; do
  echo "== libditto_$lib.so"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_$lib.so timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 4 --steps 5 2>&1 | tail -n 2
done
} > gpurun_out/r04_ab_policy.txt 2>&1
cat gpurun_out/r04_ab_policy.txt | cut -c1-220
timeout 600 python tools/lnq_bench.py > gpurun_out/r04_lnq_bench.txt 2>&1
cat gpurun_out/r04_lnq_bench.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/lnq_pmc -o p -- python3 tools/lnq_bench.py --rounds 1 --reps 5 > gpurun_out/lnq_pmc.log 2>&1
python tools/pmc_kernels.py $(find gpurun_out/lnq_pmc -name "*counter_collection.csv" | head -1) gemm_lnq > gpurun_out/r04_lnq_pmc.txt 2>&1
cat gpurun_out/r04_lnq_pmc.txt | cut -c1-260
rm -rf gpurun_out/lnq_pmc

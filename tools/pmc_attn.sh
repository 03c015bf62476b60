#!/bin/bash
# SQ counter passes over the attention microbench (one counter group per pass, no trace flags).
# usage: tools/pmc_attn.sh <attn_flags>          forward microbench (tools/attn_bench.py) at those attn_flags
#        tools/pmc_attn.sh bwd [p]               training forward + backward (tools/attn_bwd_bench.py), dropout p (default 0.0)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  if [ "$V" = bwd ]; then
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_attn_$V -o p$i -- python3 tools/attn_bwd_bench.py --p ${2:-0.0} --rounds 1 --iters 1 > gpurun_out/pmc_attn_$V.log 2>&1
  else
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_attn_$V -o p$i -- python3 tools/attn_bench.py --variants $V --rounds 1 --iters 2 > gpurun_out/pmc_attn_$V.log 2>&1
  fi
done
ls gpurun_out/pmc_attn_$V | head

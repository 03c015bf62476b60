#!/bin/bash
# A DIAGNOSTIC library that differs from the production one in ONE translation unit: compile csrc/<file>.hip with the extra -D flags
# and link it with the production objects of every other file (python -m ditto_tts_amd.build must have run).
#   tools/build_diag_one.sh libditto_diag_a2nodma.so attention.hip -DDITTO_DIAG_A2_NODMA
# Select with DITTO_HIP_LIB=$PWD/ditto_tts_amd/<name> (tools/step_ab.py, tools/gemm_bench.py).  Wrong results by design.
set -e
out=$1; file=$2; shift 2
cd "$(dirname "$0")/../ditto_tts_amd/csrc"
extra=""
[ "$file" = attention.hip ] && extra="-fno-honor-nans"
[ "$file" = attention_bwd.hip ] && extra="-fno-slp-vectorize"
{ [ "$file" = attention_train.hip ] || [ "$file" = attention_w4.hip ]; } && extra="-fno-honor-nans -fno-slp-vectorize"
obj=/tmp/diag_one_$(basename ${out%.so})_$(basename ${file%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . -Wno-unused-function -Wno-uninitialized $extra "$@" -c $file -o $obj
others=$(ls *.o | grep -v "^$(basename ${file%.hip}).o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../$out $obj $others
echo built ../$out

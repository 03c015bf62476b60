#!/bin/bash
# Build a DIAGNOSTIC copy of the library with extra -D flags (wrong results by design) next to the real one:
#   tools/build_diag.sh libditto_diag_nostore.so -DDITTO_DIAG_FAST_NOSTORE
# and select it with DITTO_HIP_LIB=$PWD/ditto_tts_amd/<name> for tools/gemm_bench.py / tools/step_ab.py.
# Flags that exist: -DDITTO_DIAG_FAST_NOSTORE (fast epilogues compute, do not store), -DDITTO_BF16_STORE_NT / _SC1 (cache policy
# of the fast epilogues' bf16 stores; measured in-model: nt 12.88 vs 12.92 ms per step = noise, sc1 worse: QKV 128 -> 140 us),
# -DDITTO_DIAG_NODMA (gemm256 main loop without its global->LDS traffic).
set -e
out=$1; shift
cd "$(dirname "$0")/../ditto_tts_amd/csrc"
mkdir -p /tmp/diag_objs
for f in *.hip ${DITTO_EXPERIMENTAL:+experimental/*.hip}; do
  extra=""; { [ "$f" = attention.hip ] || [ "$f" = experimental/attention_v4.hip ]; } && extra="-fno-honor-nans"
  [ "$f" = attention_bwd.hip ] && extra="-fno-slp-vectorize"
  { [ "$f" = attention_train.hip ] || [ "$f" = attention_w4.hip ]; } && extra="-fno-honor-nans -fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . -Wno-unused-function $extra "$@" ${DITTO_EXPERIMENTAL:+-DDITTO_EXPERIMENTAL} -c $f -o /tmp/diag_objs/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../$out /tmp/diag_objs/*.o
echo built ../$out

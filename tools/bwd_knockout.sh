#!/bin/bash
# Timing-only knock-out builds of the attention backward (csrc/attention_bwd.hip, DITTO_DIAG_BWD bits; results are WRONG by design):
#   tools/bwd_knockout.sh 1 2 4 8 16 32 48      -> ditto_tts_amd/libditto_bwdko_<bits>.so  (only attention_bwd.hip is recompiled;
# the other objects are the normal build's).  Time each with
#   DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_bwdko_<bits>.so python tools/attn_bwd_bench.py --no-check
set -e
cd "$(dirname "$0")/../ditto_tts_amd/csrc"
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . -Wno-unused-function -DDITTO_DIAG_BWD=$bits \
      -c attention_bwd.hip -o /tmp/attention_bwd_ko_$bits.o &
done
wait
for bits in "$@"; do
  objs=$(ls *.o | grep -v '^attention_bwd.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libditto_bwdko_$bits.so $objs /tmp/attention_bwd_ko_$bits.o
  echo built libditto_bwdko_$bits.so
done

#!/bin/bash
# Timing-only knock-out builds of the attention backward (csrc/attention_bwd.hip, DITTO_DIAG_BWD bits; results are WRONG by design):
#   tools/bwd_knockout.sh 1 2 4 8 16 32 48      -> ditto_tts_amd/libditto_bwdko_<bits>.so  (only attention_bwd.hip is recompiled;
# the other objects are the normal build's).  Time each with
#   DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_bwdko_<bits>.so python tools/attn_bwd_bench.py --no-check
# (under rocprofv3 --kernel-trace --stats for per-kernel times).  With DITTO_BWD_LDS_PAD=32768 one workgroup fits a CU: one wave per
# SIMD, whose timeline is serial, so the knock-outs read as additive shares.  Measured that way at C2 (dq kernel, cycles per 64-key
# tile of 2 480): MFMAs 787 (= 24 x 32: nothing hides them), P / dS vector work 822, LDS fragment reads 293, tile DMA 208.
# The knock-outs keep the surrounding work alive through opaque asm operands (a first version let the compiler delete it).
set -e
cd "$(dirname "$0")/../ditto_tts_amd/csrc"
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . -Wno-unused-function -fno-slp-vectorize -DDITTO_DIAG_BWD=$bits \
      -c attention_bwd.hip -o /tmp/attention_bwd_ko_$bits.o &
done
wait
for bits in "$@"; do
  objs=$(ls *.o | grep -v '^attention_bwd.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libditto_bwdko_$bits.so $objs /tmp/attention_bwd_ko_$bits.o
  echo built libditto_bwdko_$bits.so
done

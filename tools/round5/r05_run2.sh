#!/bin/bash
# round-5 GPU batch 2: the knock-out floors the round-4 verdict asked for (items 2, 3, 4), in the model, one box, one session:
#   attention (attn64v2) without softmax / without K/V DMA / with the exponential replaced by a plain op;
#   the fused norm2 + q-projection without its weight stream;
#   gemm256 with the next tile's K-tile-0 MFMAs issued under the fast epilogue (what progressive accumulator release could hide),
#   + the s_memtime stamp timeline of that build against the plain stamp build.
mkdir -p gpurun_out
{
echo "== shipped"; timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
for lib in a2nosm a2nodma a2expadd lnqnow epimfma; do
  echo "== libditto_diag_$lib.so"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_$lib.so timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
done
echo "== shipped (again)"; timeout 600 python tools/step_ab.py --variants "0/321~3" --rounds 3 --steps 5 2>&1 | tail -n 2
} > gpurun_out/r05_floor_diag.txt 2>&1
cut -c1-200 gpurun_out/r05_floor_diag.txt
{
echo "== stamp build"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_stamp.so timeout 300 python tools/g256_stamps.py --shapes gated,plain 2>&1 | tail -n 40
echo "== stamp build + next tile's K-tile 0 MFMAs under the epilogue"; DITTO_HIP_LIB=$PWD/ditto_tts_amd/libditto_diag_stamp_epimfma.so timeout 300 python tools/g256_stamps.py --shapes gated,plain 2>&1 | tail -n 40
} > gpurun_out/r05_g256_stamps_epimfma.txt 2>&1
grep -E "==|epilogue body|main loop|us," gpurun_out/r05_g256_stamps_epimfma.txt | cut -c1-160

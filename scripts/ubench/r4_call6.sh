#!/bin/bash
# round 4, GPU call 6: ablations of the current stream (results WRONG by design, timing only): 1 no DMA, 2 no barrier, 8 no exp / sum / pack, 16 no LDS operand reads
set -u
OUT=$PWD/gpurun_out/r4c6; mkdir -p $OUT
for shape in "256 800"; do
  for lib in seam abl1 abl2 abl8 abl16 abl17 abl27 seam abl1 abl2 abl8 abl16 abl17 abl27; do timeout 200 python scripts/ubench/pw_time.py scripts/ubench/libsavad_pw_$lib.so $shape 2>&1 | tail -1; done
done > $OUT/pw_time.log 2>&1
cat $OUT/pw_time.log

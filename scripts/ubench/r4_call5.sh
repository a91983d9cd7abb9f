#!/bin/bash
# round 4, GPU call 5: Q loads / ctx stores with the non-temporal bit (no reuse: keep them from evicting K / V^T in the L2)
set -u
OUT=$PWD/gpurun_out/r4c5; mkdir -p $OUT
for shape in "256 800" "384 800"; do
  for lib in r3 seam nt1 nt2 nt3 r3 seam nt1 nt2 nt3; do timeout 200 python scripts/ubench/pw_time.py scripts/ubench/libsavad_pw_$lib.so $shape 2>&1 | tail -1; done
done > $OUT/pw_time.log 2>&1
cat $OUT/pw_time.log
for v in seam nt3 nt1 nt2; do timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_$v.so r4c5_$v > $OUT/pmc_$v.log 2>&1; grep -E "attention_pw" $OUT/pmc_$v.log; done

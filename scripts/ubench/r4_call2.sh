#!/bin/bash
# round 4, GPU call 2: key-split item with scores / PV on alternating stages and its DMA pieces in the MFMA gaps
set -u
OUT=gpurun_out/r4c2; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q -k "persistent or automatic_picks or config4" 2>&1 | tail -25 ) > $OUT/pytest_pw.log
tail -5 $OUT/pytest_pw.log
for shape in "256 800" "384 800" "256 801"; do
  for lib in r3 ks1 ks2 ks2nobar r3 ks1 ks2; do timeout 200 python scripts/ubench/pw_time.py scripts/ubench/libsavad_pw_$lib.so $shape 2>&1 | tail -1; done
done > $OUT/pw_time.log 2>&1
cat $OUT/pw_time.log
timeout 200 python scripts/ubench/pw_timing.py 256 800 > $OUT/pw_timing.log 2>&1; cat $OUT/pw_timing.log
timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_ks2.so r4c2_ks2 > $OUT/pmc_ks2.log 2>&1; cat $OUT/pmc_ks2.log
timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_r3.so r4c2_r3 > $OUT/pmc_r3.log 2>&1; cat $OUT/pmc_r3.log

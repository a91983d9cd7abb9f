#!/bin/bash
# round 4, GPU call 1: key-split tail item + attached order -- correctness, same-box A/B against round 3's stream, stamps, traffic
set -u
OUT=gpurun_out/r4c1; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q -k "persistent or automatic_picks" 2>&1 | tail -25 ) > $OUT/pytest_pw.log
tail -5 $OUT/pytest_pw.log
for shape in "256 800" "384 800" "256 801" "300 1000"; do
  for lib in r3 ks ksna ksnobar r3 ks; do timeout 200 python scripts/ubench/pw_time.py scripts/ubench/libsavad_pw_$lib.so $shape 2>&1 | tail -1; done
done > $OUT/pw_time.log 2>&1
cat $OUT/pw_time.log
timeout 200 python scripts/ubench/pw_timing.py 256 800 > $OUT/pw_timing.log 2>&1; cat $OUT/pw_timing.log
timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_ks.so r4c1_ks > $OUT/pmc_ks.log 2>&1; cat $OUT/pmc_ks.log
timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_ksna.so r4c1_ksna > $OUT/pmc_ksna.log 2>&1; cat $OUT/pmc_ksna.log
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_all.log
tail -8 $OUT/pytest_all.log

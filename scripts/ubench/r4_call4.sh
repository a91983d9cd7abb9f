#!/bin/bash
# round 4, GPU call 4: balanced key-split stages (ks3) and the overlapped item seam (seam) against ks2 / round 3, same box
set -u
OUT=$PWD/gpurun_out/r4c4; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q -k "persistent or automatic_picks or config4 or bf16" 2>&1 | tail -25 ) > $OUT/pytest_pw.log
tail -5 $OUT/pytest_pw.log
for shape in "256 800" "384 800" "256 801" "300 1000"; do
  for lib in r3 ks2 ks3 seam r3 ks2 ks3 seam; do timeout 200 python scripts/ubench/pw_time.py scripts/ubench/libsavad_pw_$lib.so $shape 2>&1 | tail -1; done
done > $OUT/pw_time.log 2>&1
cat $OUT/pw_time.log
timeout 200 python scripts/ubench/pw_timing.py 256 800 > $OUT/pw_timing.log 2>&1; cat $OUT/pw_timing.log
cd /tmp; rocprofv3 --list-avail 2>/dev/null > $OUT/counters_all.txt; grep -iE "MALL|HBM|DRAM|UMC|EA_RD|EA_WR|TCC_EA|TCC_REQ|TCC_HIT|TCC_MISS" $OUT/counters_all.txt | head -60

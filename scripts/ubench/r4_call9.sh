#!/bin/bash
# round 4, GPU call 9: cache policies of the Q loads / ctx stores (sc0 sc1: system scope), timing + fabric traffic
set -u
OUT=$PWD/gpurun_out/r4c9; mkdir -p $OUT
for lib in cur q2c2 q3c3 q2 c2 q4c4 cur q2c2 q3c3 q2 c2 q4c4; do timeout 200 python scripts/ubench/pw_time.py scripts/ubench/libsavad_pw_$lib.so 256 800 2>&1 | tail -1; done | tee $OUT/pw_time.log
for v in cur q2c2 q3c3 q4c4; do timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_$v.so r4c9_$v > $OUT/pmc_$v.log 2>&1; echo $v; grep -E "attention_pw.*(FETCH|WRITE|avg_ns)" $OUT/pmc_$v.log; done
timeout 100 python scripts/ubench/pw_check.py 4 800 3 264 2>&1 | tail -3

#!/bin/bash
# round 4, GPU call 10: where do the extra 113 MB of fabric reads come from?  T = 768 has no tail group (QB = 24): same kernel, no key-split item
set -u
OUT=$PWD/gpurun_out/r4c10; mkdir -p $OUT
for spec in "cur 256 768" "r3 256 768" "cur 256 800" "cur 512 800" "cur 128 1600"; do set -- $spec
  timeout 600 bash scripts/ubench/pw_pmc.sh scripts/ubench/libsavad_pw_$1.so r4c10_$1_$2_$3 $2 $3 > $OUT/pmc_$1_$2_$3.log 2>&1; echo "$spec"; grep -E "attention_pw.*(FETCH|WRITE|avg_ns)" $OUT/pmc_$1_$2_$3.log; done

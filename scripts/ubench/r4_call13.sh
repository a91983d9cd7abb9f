#!/bin/bash
# round 4, GPU call 13: the round's profiles again (the first attempt lost its output), every rocprofv3 pass under its own timeout
set -u
OUT=$PWD/gpurun_out/r4c13; mkdir -p $OUT
t0=$(date +%s)
timeout 420 bash scripts/profile_gpu.sh r4_bf16 --precision bf16 --batch 256 --no-secondary > $OUT/prof_r4_bf16.log 2>&1; echo "bf16 done $(( $(date +%s) - t0 )) s"; grep "rc=" $OUT/prof_r4_bf16.log | tr '\n' ' '
timeout 500 bash scripts/profile_gpu.sh r4 > $OUT/prof_r4.log 2>&1; echo "fp32 done $(( $(date +%s) - t0 )) s"; grep "rc=" $OUT/prof_r4.log | tr '\n' ' '
if [ $(( $(date +%s) - t0 )) -lt 560 ]; then timeout 260 bash scripts/profile_gpu.sh r4_t7 --batch 1000 --frames 7 --no-secondary > $OUT/prof_r4_t7.log 2>&1; echo "t7 done $(( $(date +%s) - t0 )) s"; grep "rc=" $OUT/prof_r4_t7.log | tr '\n' ' '; fi
find gpurun_out -name "*.csv" -size +1M -delete
du -sh gpurun_out

#!/bin/bash
# round 4, GPU call 12: whole GPU suite, then the round's profiles (rocprofv3 trace + PMC passes) for the four profiled workloads
set -u
OUT=$PWD/gpurun_out/r4c12; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest_all.log; tail -8 $OUT/pytest_all.log
timeout 900 bash scripts/profile_gpu.sh r4 > $OUT/prof_r4.log 2>&1; tail -3 $OUT/prof_r4.log
timeout 900 bash scripts/profile_gpu.sh r4_bf16 --precision bf16 --batch 256 > $OUT/prof_r4_bf16.log 2>&1; tail -3 $OUT/prof_r4_bf16.log
timeout 600 bash scripts/profile_gpu.sh r4_t7 --batch 1000 --frames 7 > $OUT/prof_r4_t7.log 2>&1; tail -3 $OUT/prof_r4_t7.log
timeout 600 bash scripts/profile_gpu.sh r4_t50 --batch 512 --frames 50 > $OUT/prof_r4_t50.log 2>&1; tail -3 $OUT/prof_r4_t50.log
ls gpurun_out/prof_r4*/

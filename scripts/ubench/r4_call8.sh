#!/bin/bash
# round 4, GPU call 8: committed sources -- whole GPU suite, sweep of both attention kernels with the balanced tail attachment, host overhead
set -u
OUT=$PWD/gpurun_out/r4c8; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_all.log; tail -6 $OUT/pytest_all.log
timeout 600 python scripts/ubench/pw_sweep.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pw_sweep.log
timeout 200 python scripts/ubench/host_overhead.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_overhead.log
timeout 200 python scripts/ubench/pw_timing.py 256 800 2>&1 | grep -v amdgpu.ids > $OUT/pw_timing.log; cat $OUT/pw_timing.log

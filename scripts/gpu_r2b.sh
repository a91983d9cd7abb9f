#!/bin/bash
set -u
OUT=gpurun_out/r2b; mkdir -p $OUT
for v in base abl1 abl2 abl4 abl8 abl15; do
  python scripts/ubench/attn2_probe.py scripts/ubench/libsavad_$v.so 6 256 >> $OUT/probe.log 2>&1
done
python scripts/ubench/attn2_probe.py scripts/ubench/libsavad_base.so 1 256 >> $OUT/probe.log 2>&1
python scripts/ubench/attn2_probe.py scripts/ubench/libsavad_timing.so 6 256 t >> $OUT/probe.log 2>&1
( SAVAD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 ) > $OUT/bench_dist1.log
cat $OUT/probe.log

#!/bin/bash
set -u
OUT=gpurun_out/r2c; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16" 2>&1 | tail -15 ) > $OUT/pytest_bf16.log
for v in base abl1 abl8 abl11; do
  python scripts/ubench/attn2_probe.py scripts/ubench/libsavad_$v.so 6 256 >> $OUT/probe.log 2>&1
done
python scripts/ubench/attn2_probe.py scripts/ubench/libsavad_base.so 1 256 >> $OUT/probe.log 2>&1
python scripts/ubench/attn2_probe.py scripts/ubench/libsavad_timing.so 6 256 t >> $OUT/probe.log 2>&1
cat $OUT/pytest_bf16.log; grep -v amdgpu.ids $OUT/probe.log

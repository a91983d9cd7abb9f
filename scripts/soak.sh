#!/bin/bash
# Robustness soak on the GPU box with whatever budget is left ($1 seconds): the cache-pressure file, the multi-stream RCCL file and the
# randomised parity sweep, alternating, until the budget is spent -> gpurun_out/soak.log (one line per pass)
BUDGET=${1:-300}; T0=$(date +%s); OUT=gpurun_out/soak.log; mkdir -p gpurun_out; : > $OUT
python -c "import sys; sys.path.insert(0, '.'); import bench; print('csrc_hash', bench.kernel_source_hash())" >> $OUT
n=0
while [ $(( $(date +%s) - T0 + 40 )) -lt $BUDGET ]; do
  timeout 120 python -m pytest tests/test_gpu_cache_pressure.py -q 2>&1 | grep -E "passed|failed" | sed "s/^/cache_pressure: /" >> $OUT
  timeout 60 python -m pytest tests -m gpu -q -k dist_nccl 2>&1 | grep -E "passed|failed" | sed "s/^/rccl: /" >> $OUT
  timeout 120 python scripts/fuzz_parity.py $((${SEED0:-1000} + n)) 40 2>&1 | tail -1 | sed "s/^/fuzz seed $((${SEED0:-1000} + n)): /" >> $OUT
  n=$((n + 1))
done
echo "passes: $n, $(grep -cE "[0-9]+ failed|error" $OUT) lines with failures, $(( $(date +%s) - T0 )) s" >> $OUT
cat $OUT

#!/bin/bash
# Runs ON the GPU box: kernel traces of the default bench (batches in flight) and of --in-flight 1, reduced to scripts/overlap_from_trace.py's
# summary (the raw traces stay on the box).  Output: gpurun_out/overlap_<tag>.txt
TAG=${1:-r3}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/overlap_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $REPO/gpurun_out/overlap_$TAG.txt
for mode in 0 1; do
  rm -rf $OUT/t$mode
  rocprofv3 --kernel-trace -f csv -d $OUT/t$mode -o trace -- python $REPO/bench.py --in-flight $mode --no-cpu-baseline --legs none --no-events --min-seconds 0.3 "$@" > $OUT/t$mode.log 2>&1
  f=$(find $OUT/t$mode -name "*kernel_trace.csv" | head -1)
  python $REPO/scripts/overlap_from_trace.py $f "bench.py --in-flight $mode $*" >> $REPO/gpurun_out/overlap_$TAG.txt 2>&1
  grep '^{"metric"' $OUT/t$mode.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench line under the tracer: in_flight', d['in_flight'], 'ms_per_step', d['ms_per_step'], 'value', d['value'])" >> $REPO/gpurun_out/overlap_$TAG.txt 2>&1
  rm -rf $OUT/t$mode
done
cat $REPO/gpurun_out/overlap_$TAG.txt

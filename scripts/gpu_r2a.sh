#!/bin/bash
# Round-2 GPU call A: full GPU test suite, the new bench line, bf16 attention generations side by side.
set -u
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest.log
( timeout 300 python bench.py 2>&1 | tail -3 ) > $OUT/bench_default.log
( timeout 200 python bench.py --precision bf16 --batch 256 --row-mode 1 --no-cpu-baseline --no-secondary 2>&1 | tail -2 ) > $OUT/bench_bf16_mode1.log
( timeout 200 python bench.py --precision bf16 --batch 256 --row-mode 6 --no-cpu-baseline --no-secondary 2>&1 | tail -2 ) > $OUT/bench_bf16_mode6.log
( SAVAD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -3 ) > $OUT/bench_dist1.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/trace6 -o trace -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --batch 256 --row-mode 6 --no-cpu-baseline --no-secondary --no-events --min-seconds 0.1 > $GRAFT_REPO_ROOT/$OUT/trace6.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*.csv" -size +2M -delete
head -c 3000 $OUT/pytest.log; echo; cat $OUT/bench_bf16_mode1.log | head -c 1500; echo; cat $OUT/bench_bf16_mode6.log | head -c 1500

#!/bin/bash
# One GPU call that produces everything a round commits: the GPU test suite, the four profile sets (scripts/profile_gpu.sh),
# the bench lines (taken AFTER the fresh profiles were copied into profiles/ on the box, so that their traffic fields carry this
# tree's csrc_hash), and -- with what is left of BUDGET seconds -- repeated runs of the multi-stream RCCL tests (the ones that
# caught the registers-with-a-load-in-flight race of round 4).  Usage: [SKIP_TESTS=1] scripts/final_measurements.sh [BUDGET seconds, default 600]
set -u
BUDGET=${1:-600}
T0=$(date +%s)
OUT=$PWD/gpurun_out/r4final; mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != 1 ]; then
    ( timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) | tee $OUT/pytest.log
    # (the summary is not always the last line: RCCL's banner can follow it on stderr)
    if ! grep -qE "^[0-9]+ passed" $OUT/pytest.log || grep -qE "^[0-9]+ (failed|error)| [0-9]+ (failed|error)" $OUT/pytest.log; then
        echo "GPU tests did not pass: nothing else is run"; timeout 200 python -m pytest tests -m gpu -q -x 2>&1 | tail -60 > $OUT/pytest_fail.log; exit 1
    fi
fi
timeout 100 bash scripts/profile_gpu.sh r4_bf16 --precision bf16 --batch 256 --no-secondary > $OUT/prof_r4_bf16.log 2>&1; grep "rc=" $OUT/prof_r4_bf16.log | tr '\n' ' '
timeout 100 bash scripts/profile_gpu.sh r4 > $OUT/prof_r4.log 2>&1; grep "rc=" $OUT/prof_r4.log | tr '\n' ' '
timeout 100 bash scripts/profile_gpu.sh r4_t7 --batch 1000 --frames 7 --no-secondary > $OUT/prof_r4_t7.log 2>&1; grep "rc=" $OUT/prof_r4_t7.log | tr '\n' ' '
timeout 100 bash scripts/profile_gpu.sh r4_t50 --batch 512 --frames 50 --no-secondary > $OUT/prof_r4_t50.log 2>&1; grep "rc=" $OUT/prof_r4_t50.log | tr '\n' ' '
for t in "r4_bf16 r4_bf16_b256" r4 r4_t7 r4_t50; do bash scripts/collect_profiles.sh $t > /dev/null 2>&1 || echo "collect $t failed"; done
( timeout 200 python bench.py 2>$OUT/bench.err | grep "^{" ) > $OUT/bench.json; wc -c $OUT/bench.json
( SAVAD_BENCH_FORCE_DIST=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep "^{" ) > $OUT/bench_dist1.json; wc -c $OUT/bench_dist1.json
find gpurun_out -name "*.csv" -size +1M -delete
echo "measurements done after $(( $(date +%s) - T0 )) s"
n=0
while [ $(( $(date +%s) - T0 + 12 )) -lt $BUDGET ]; do
    timeout 60 python -m pytest tests -m gpu -x -q -k dist_nccl 2>&1 | grep -E "^FAILED|passed|failed|rows differ" | tail -3 | tee -a $OUT/repeat.log
    n=$((n + 1))
done
echo "repeated the RCCL test file $n times: $(grep -c ' passed' $OUT/repeat.log 2>/dev/null) clean, $(grep -c 'failed' $OUT/repeat.log 2>/dev/null) with failures" | tee -a $OUT/repeat.log

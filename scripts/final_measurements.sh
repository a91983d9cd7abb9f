#!/bin/bash
# One GPU call that produces everything a round commits: the GPU test suite, the profile sets (scripts/profile_gpu.sh),
# the bench lines (taken AFTER the fresh profiles were copied into profiles/ on the box, so that their traffic fields carry this
# tree's csrc_hash), and -- with what is left of BUDGET seconds -- repeated runs of the multi-stream RCCL tests (the ones that
# caught the registers-with-a-load-in-flight race of round 4).  Usage: [SKIP_TESTS=1] [R=r6] scripts/final_measurements.sh [BUDGET seconds, default 60]
set -u
BUDGET=${1:-60}
R=${R:-r6}
T0=$(date +%s)
OUT=$PWD/gpurun_out/${R}final; mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != 1 ]; then
    ( timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) | tee $OUT/pytest.log
    # (the summary is not always the last line: RCCL's banner can follow it on stderr)
    if ! grep -qE "^[0-9]+ passed" $OUT/pytest.log || grep -qE "^[0-9]+ (failed|error)| [0-9]+ (failed|error)" $OUT/pytest.log; then
        echo "GPU tests did not pass: nothing else is run"; timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -60 > $OUT/pytest_fail.log; exit 1
    fi
fi
prof() { tag=$1; shift; timeout 170 bash scripts/profile_gpu.sh $tag "$@" > $OUT/prof_$tag.log 2>&1; echo "$tag: $(grep 'rc=' $OUT/prof_$tag.log | tr '\n' ' ')"; }
prof ${R}_fp32s                                                                   # the headline: configs[1] [32,800,80], split-bf16 operands
prof ${R} --precision fp32                                                        # the same config on the exact-fp32 MFMA (rounds 1-5's headline)
prof ${R}_bf16 --precision bf16 --batch 256 --no-secondary                        # configs[2]
prof ${R}_t7 --precision fp32 --batch 1000 --frames 7 --no-secondary              # the reference's window batch, exact fp32
prof ${R}_t7_fp32s --precision fp32s --batch 1000 --frames 7 --no-secondary       # ... fp32s: the latency variant of the single launch
prof ${R}_t7_fp32s_big --precision fp32s --batch 65536 --frames 7 --no-secondary --steps 4   # ... a wave per block
prof ${R}_t7_bf16 --precision bf16 --batch 1000 --frames 7 --no-secondary
prof ${R}_t7_bf16_big --precision bf16 --batch 65536 --frames 7 --no-secondary --steps 4
timeout 150 bash scripts/profile_logmel.sh ${R}_logmel > $OUT/prof_${R}_logmel.log 2>&1; grep "rc=" $OUT/prof_${R}_logmel.log | tr '\n' ' '
for t in "${R}_bf16 ${R}_bf16_b256" ${R} ${R}_fp32s ${R}_t7 ${R}_t7_fp32s ${R}_t7_fp32s_big ${R}_t7_bf16 ${R}_t7_bf16_big ${R}_logmel; do bash scripts/collect_profiles.sh $t > /dev/null 2>&1 || echo "collect $t failed"; done
timeout 200 bash scripts/profile_overlap.sh ${R} > $OUT/overlap.log 2>&1; cp gpurun_out/overlap_${R}.txt profiles/${R}_inflight_overlap.txt 2>/dev/null
( timeout 300 python bench.py 2>$OUT/bench.err | grep "^{" ) > $OUT/bench.json; wc -c $OUT/bench.json
( SAVAD_BENCH_FORCE_DIST=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep "^{" ) > $OUT/bench_dist1.json; wc -c $OUT/bench_dist1.json
find gpurun_out -name "*.csv" -size +1M -delete
echo "measurements done after $(( $(date +%s) - T0 )) s"
n=0
while [ $(( $(date +%s) - T0 + 12 )) -lt $BUDGET ]; do
    timeout 60 python -m pytest tests -m gpu -x -q -k dist_nccl 2>&1 | grep -E "^FAILED|passed|failed|rows differ" | tail -3 | tee -a $OUT/repeat.log
    n=$((n + 1))
done
echo "repeated the RCCL test file $n times: $(grep -c ' passed' $OUT/repeat.log 2>/dev/null) clean, $(grep -c 'failed' $OUT/repeat.log 2>/dev/null) with failures" | tee -a $OUT/repeat.log

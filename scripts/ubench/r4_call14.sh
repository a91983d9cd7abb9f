#!/bin/bash
# round 4, GPU call 14: the bench line of the round on the committed sources (+ the ragged-batch profile, + the RCCL world-1 line)
set -u
OUT=$PWD/gpurun_out/r4c14; mkdir -p $OUT
( timeout 600 python bench.py 2>$OUT/bench.err | grep "^{" ) > $OUT/bench.json; wc -c $OUT/bench.json
( SAVAD_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep "^{" ) > $OUT/bench_dist1.json; wc -c $OUT/bench_dist1.json
timeout 300 bash scripts/profile_gpu.sh r4_t50 --batch 512 --frames 50 --no-secondary > $OUT/prof_r4_t50.log 2>&1; grep "rc=" $OUT/prof_r4_t50.log | tr '\n' ' '
find gpurun_out -name "*.csv" -size +1M -delete

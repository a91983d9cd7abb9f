#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4final; mkdir -p $OUT
( timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) | tee $OUT/pytest.log
timeout 300 bash scripts/profile_gpu.sh r4_bf16 --precision bf16 --batch 256 --no-secondary > $OUT/prof_r4_bf16.log 2>&1; grep "rc=" $OUT/prof_r4_bf16.log | tr '\n' ' '
timeout 400 bash scripts/profile_gpu.sh r4 > $OUT/prof_r4.log 2>&1; grep "rc=" $OUT/prof_r4.log | tr '\n' ' '
timeout 260 bash scripts/profile_gpu.sh r4_t7 --batch 1000 --frames 7 --no-secondary > $OUT/prof_r4_t7.log 2>&1; grep "rc=" $OUT/prof_r4_t7.log | tr '\n' ' '
timeout 260 bash scripts/profile_gpu.sh r4_t50 --batch 512 --frames 50 --no-secondary > $OUT/prof_r4_t50.log 2>&1; grep "rc=" $OUT/prof_r4_t50.log | tr '\n' ' '
( timeout 600 python bench.py 2>$OUT/bench.err | grep "^{" ) > $OUT/bench.json; wc -c $OUT/bench.json
( SAVAD_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep "^{" ) > $OUT/bench_dist1.json; wc -c $OUT/bench_dist1.json
find gpurun_out -name "*.csv" -size +1M -delete

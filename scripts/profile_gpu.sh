#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel trace + PMC passes of the default bench workload.
# Usage: scripts/profile_gpu.sh <tag> [extra bench args]
set -u
TAG=${1:-r1}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# --in-flight 1: one forward at a time, so that a launch's duration (trace) and its counters (PMC) are the kernel's own -- bench.py's
# default keeps up to three batches in flight, whose launches overlap
BENCH="python $REPO/bench.py --in-flight 1 --steps 10 --warmup 3 --min-seconds 0.02 --no-secondary --no-cpu-baseline $*"
# the kernel-trace pass runs the SAME command as the default bench line (K = 50, W = 10, secondary legs included): its
# per-kernel averages are the ones bench.py's HIP-event durations are checked against; the PMC passes serialise kernels,
# so they run the primary workload only and stay short
# (the legs that run the profiled kernels at ONE shape each: their averages feed bench.py's event_over_rocprof)
TRACE_BENCH="python $REPO/bench.py --in-flight 1 --no-cpu-baseline --legs configs2_bf16_b256_t800,pipeline_fp32_b1000_t7,configs0_clip10s_audio_to_probabilities $*"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $TRACE_BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
echo "pmc_sq rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_l2 -o pmc -- $BENCH > $OUT/pmc_l2.log 2>&1
echo "pmc_l2 rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 -f csv -d $OUT/pmc_inst -o pmc -- $BENCH > $OUT/pmc_inst.log 2>&1
echo "pmc_inst rc=$?"
# (round 4: a pass over the L2 -> fabric request counters -- TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
#  TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum -- never returned on this stack: rocprofv3 sat until the pass's timeout, four workloads in a
#  row, 40 GPU-minutes; every pass now runs under its own timeout and that one is not attempted)
cd $REPO
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merged-back payload small
find $OUT -name "*.csv" -size +4M -delete

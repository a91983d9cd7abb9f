#!/bin/bash
# Runs ON the GPU box: kernel trace + PMC passes of the log-mel front-end on an hour of audio (scripts/ubench/logmel_bench.py --one).
# Usage: scripts/profile_logmel.sh <tag>      -> gpurun_out/prof_<tag>/{summary.txt,kernel_avg.json,traffic.json}
set -u
TAG=${1:-r5_logmel}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $REPO/scripts/ubench/logmel_bench.py --one"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1; echo "pmc_write rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 -f csv -d $OUT/pmc_inst -o pmc -- $CMD > $OUT/pmc_inst.log 2>&1; echo "pmc_inst rc=$?"
cd $REPO
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
find $OUT -name "*.csv" -size +4M -delete

#!/bin/bash
# HBM-side traffic + kernel trace of the bf16 forward at [B, T] with one library variant (runs ON the GPU box):
#   pw_pmc.sh <lib.so> <tag> [B T]   -> gpurun_out/prof_<tag>/{summary.txt,traffic.json,kernel_avg.json}
set -u
LIB=$(realpath $1); TAG=$2; B=${3:-256}; T=${4:-800}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/scripts/ubench/pw_time.py $LIB $B $T 0 12"
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/scripts/ubench/pw_time.py $LIB $B $T > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1; echo "pmc_write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
cd $REPO
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
grep -E "attention|row_kernel_bf16|input_qkv" $OUT/summary.txt | head -40
find $OUT -name "*.csv" -size +2M -delete

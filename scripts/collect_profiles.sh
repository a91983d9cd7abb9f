#!/bin/bash
# Copies what scripts/profile_gpu.sh <tag> left under gpurun_out/prof_<tag> into the tracked profiles/ directory:
#   summary.txt -> <tag>_rocprofv3_summary.txt, the kernel-trace statistics -> <tag>_kernel_stats.csv,
#   kernel_avg.json -> <tag>_kernel_avg.json, traffic.json -> <dst>_traffic.json (dst defaults to tag).
# Usage: scripts/collect_profiles.sh <tag> [traffic-name]
set -e
cd "$(dirname "$0")/.."
TAG=$1; DST=${2:-$1}
SRC=gpurun_out/prof_$TAG
cp $SRC/summary.txt profiles/${TAG}_rocprofv3_summary.txt
cp "$(ls $SRC/trace/*/*kernel_stats.csv $SRC/trace/*kernel_stats.csv 2>/dev/null | head -1)" profiles/${TAG}_kernel_stats.csv
cp $SRC/kernel_avg.json profiles/${TAG}_kernel_avg.json
cp $SRC/traffic.json profiles/${DST}_traffic.json
ls -la profiles/${TAG}_* profiles/${DST}_traffic.json

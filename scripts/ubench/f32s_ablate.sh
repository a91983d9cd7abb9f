#!/bin/bash
# fp32s fused launch, where the time goes: experiment builds with pieces switched off (SAVAD_ABLATE bits in csrc/savad_kernels_f32s.h:
# 1 no DMA, 2 no ring wait / barrier, 4 no softmax, 8 no LDS operand reads, 128 no split arithmetic), timed on the same box.
#   build (CPU):  scripts/ubench/f32s_ablate.sh build "1 2 3 4 8 128 132 143"
#   run (GPU):    scripts/ubench/f32s_ablate.sh run "1 2 3 4 8 128 132 143" [B T]
set -e
cd "$(dirname "$0")/../.."
mode=$1; variants=$2; B=${3:-32}; T=${4:-800}
if [ "$mode" = build ]; then
  for v in $variants; do
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -w -DSAVAD_ABLATE=$v \
        voice_activity_detection_amd/csrc/savad.hip -o scripts/ubench/libsavad_abl_$v.so ) &
    while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 1; done
  done
  wait
else
  SAVAD_LIB=$PWD/voice_activity_detection_amd/libsavad.so python scripts/ubench/ab_lib.py --one fp32s $B $T
  for v in $variants; do
    SAVAD_LIB=$PWD/scripts/ubench/libsavad_abl_$v.so timeout 300 python scripts/ubench/ab_lib.py --one fp32s $B $T
  done
  SAVAD_LIB=$PWD/voice_activity_detection_amd/libsavad.so python scripts/ubench/ab_lib.py --one fp32s $B $T
fi

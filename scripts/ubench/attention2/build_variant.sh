#!/bin/bash
# usage: build_variant.sh <header under scripts/ubench/attention2> <out name> [extra -D flags]
cd "$(dirname "$0")/../../.."
H=$(pwd)/scripts/ubench/attention2/$1; OUT=scripts/ubench/libsavad_$2.so; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -Wno-unused-value -DSAVAD_ATTN2="\"$H\"" "$@" \
  -Ivoice_activity_detection_amd/csrc voice_activity_detection_amd/csrc/savad.hip -o $OUT 2>&1 | grep -E "error|warning: inline" | head -5
ls -la $OUT | awk '{print $5, $9}'

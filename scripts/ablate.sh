#!/bin/bash
# Build ablated variants of libsavad.so (timing experiments only; results are WRONG by design).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for a in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -Wno-unused-value -DSAVAD_ABLATE=$a \
     voice_activity_detection_amd/csrc/savad.hip -o scripts/ubench/libsavad_ablate$a.so
done

#!/bin/bash
# timing build of the library (phase stamps of the persistent kernels): scripts/ubench/libsavad_timing.so
cd "$(dirname "$0")/../.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -w -DSAVAD_TIMING "$@" \
   voice_activity_detection_amd/csrc/savad.hip -o scripts/ubench/libsavad_timing.so && ls -la scripts/ubench/libsavad_timing.so

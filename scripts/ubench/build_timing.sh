#!/bin/bash
# timing build of the library (phase stamps of the persistent attention kernel): scripts/ubench/libsavad_timing.so
cd "$(dirname "$0")/../.."
mkdir -p scripts/ubench/gen
python scripts/gen_attn_pw.py --timing --out scripts/ubench/gen/savad_attn_pw_bf16_timing.inc || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -w -DSAVAD_TIMING \
   -DSAVAD_PW_INC="\"$PWD/scripts/ubench/gen/savad_attn_pw_bf16_timing.inc\"" "$@" \
   voice_activity_detection_amd/csrc/savad.hip -o scripts/ubench/libsavad_timing.so && ls -la scripts/ubench/libsavad_timing.so

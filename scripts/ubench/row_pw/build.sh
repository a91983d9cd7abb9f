#!/bin/bash
# Builds the row_pw experiment: libsavad_rowpw.so (+ libsavad_rowpw_timing.so with phase stamps) next to this script.
# The product sources are not touched: a patched COPY of savad.hip gets the row_mode 7 launch hook (savad_hip_hooks.diff).
set -e
cd "$(dirname "$0")"
REPO=$(cd ../../.. && pwd)
python gen_row_pw.py
mkdir -p gen
cp "$REPO/voice_activity_detection_amd/csrc/savad.hip" gen/savad_rowpw.hip
patch -s gen/savad_rowpw.hip savad_hip_hooks.diff
for v in "" "_timing"; do
  flags=""; [ -n "$v" ] && flags="-DSAVAD_TIMING"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -w $flags -I"$REPO/voice_activity_detection_amd/csrc" -I. -Igen \
     gen/savad_rowpw.hip -o libsavad_rowpw$v.so &
done
wait
ls -la libsavad_rowpw*.so

#!/bin/bash
# scripts/ubench/abl/libsavad_head.so = the library of commit ${1:-HEAD}'s sources (for ab_head.sh)
rm -rf /tmp/savad_head && mkdir -p /tmp/savad_head && git archive ${1:-HEAD} voice_activity_detection_amd/csrc include | tar -x -C /tmp/savad_head &&
(cd /tmp/savad_head && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -Wno-unused-value voice_activity_detection_amd/csrc/savad.hip -o "$OLDPWD/scripts/ubench/abl/libsavad_head.so")

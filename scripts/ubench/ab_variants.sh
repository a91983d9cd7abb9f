#!/bin/bash
# same-run A/B of whole-library compiler-option variants (GPU box).  Build them first, e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -w -mllvm -amdgpu-sched-strategy=max-ilp \
#         voice_activity_detection_amd/csrc/savad.hip -o scripts/ubench/libsavad_v_ilp.so
# Round 3 (ms per forward, product / variant): fp32 [32,800]: max-ilp 0.6290 / 0.6303, max-memory-clause 0.6291 / 0.6273,
# metric-bias 0 0.6294 / 0.6304, metric-bias 100 equal, amdgpu-trackers 0.6315 / 0.6328; bf16 [256,800]: max-ilp 0.6148 / 0.6323,
# max-memory-clause 0.6138 / 0.6422, bias 0 and trackers equal.  launch_bounds(256,1) on the fused fp32 kernel: 0.6258 / 0.6467.
# Non-temporal stores (`__builtin_nontemporal_store`) for the bf16 q / k / v^T / residual outputs: 0.6101 / 0.6306.
# s_setprio 1 / 3 around the ring GEMMs of the bf16 row chain (0 elsewhere): 0.6069 / 0.6210 and 0.6085 / 0.6202.
# Nothing to adopt.
for v in ilp memclause bias0 bias100 trackers; do
  timeout 120 python scripts/ubench/ab_lib.py voice_activity_detection_amd/libsavad.so scripts/ubench/libsavad_v_$v.so fp32 32 800 1
done
for v in ilp memclause bias0 trackers; do
  timeout 120 python scripts/ubench/ab_lib.py voice_activity_detection_amd/libsavad.so scripts/ubench/libsavad_v_$v.so bf16 256 800 1
done

#!/bin/bash
# same-box A/B of the working tree's library against scripts/ubench/abl/libsavad_head.so (scripts/ubench/build_head.sh: an earlier
# commit's sources): a bench leg with its per-kernel block; "$@" = extra bench arguments
for i in 1 2; do for lib in scripts/ubench/abl/libsavad_head.so voice_activity_detection_amd/libsavad.so; do
SAVAD_LIB=$PWD/$lib timeout 200 python bench.py --no-secondary --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline'].get('per_kernel', {}); print('$lib'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_one_in_flight'), {k:v['ms'] for k,v in pk.items()})"
done; done

#!/bin/bash
# same-box A/B of the working tree's library against scripts/ubench/abl/libsavad_head.so (the sources of HEAD, built by hand):
# the headline, the bf16 [256,800,80] leg, the T = 7 legs
for i in 1 2; do for lib in scripts/ubench/abl/libsavad_head.so voice_activity_detection_amd/libsavad.so; do
SAVAD_LIB=$PWD/$lib timeout 200 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']; print('$lib'.split('/')[-1], 'fp32', d['ms_per_step'], d.get('ms_per_step_one_in_flight'), {k:v['ms'] for k,v in pk.items()})"
SAVAD_LIB=$PWD/$lib timeout 200 python bench.py --precision bf16 --batch 65536 --frames 7 --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], 'bf16 [65536,7]', d['ms_per_step'], d.get('ms_per_step_one_in_flight'))"
SAVAD_LIB=$PWD/$lib timeout 200 python bench.py --precision bf16 --batch 4096 --frames 7 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], 'bf16 [4096,7]', d['ms_per_step'], d.get('ms_per_step_one_in_flight'))"
done; done

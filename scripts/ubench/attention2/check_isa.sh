#!/bin/bash
# usage: check_isa.sh <header> [-D flags]: resource usage + compiler-generated AGPR uses of attention2_kernel_bf16
H=/root/repo/scripts/ubench/attention2/$1; shift
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DSAVAD_ATTN2="\"$H\"" "$@" -I/root/repo/voice_activity_detection_amd/csrc -o chk.s /root/repo/voice_activity_detection_amd/csrc/savad.hip 2>&1 | grep error
grep "attention2" -A14 chk.s | grep "NumVgprs\|ScratchSize\|NumAgprs\|NumSgprs"
awk '/^_ZN5savad2bf22attention2_kernel_bf16/,/s_endpgm/' chk.s > chk_a2.s
python3 - <<'PY'
import re
inasm=False; n=0
for l in open('/tmp/isa/chk_a2.s'):
    if '#ASMSTART' in l: inasm=True; continue
    if '#ASMEND' in l: inasm=False; continue
    if not inasm and ('accvgpr' in l or re.search(r'\ba\d+\b|\ba\[',l.split(';')[0])): n+=1
print('compiler-generated AGPR uses outside asm:',n, '(must be 0)')
PY

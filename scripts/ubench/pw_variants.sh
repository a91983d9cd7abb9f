#!/bin/bash
# build experiment variants of the persistent attention stream: pw_variants.sh "<name> <gen args>" ...
# -> scripts/ubench/libsavad_pw_<name>.so   (results of ablated variants are WRONG by design; timing only)
# a spec "<name> @<file.inc>" takes an existing instruction stream instead of generating one
cd "$(dirname "$0")/../.."
mkdir -p /tmp/pwv
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  if [[ "${1:-}" == @* ]]; then
    cp "${1#@}" /tmp/pwv/$name.inc || exit 1
  else
    python scripts/gen_attn_pw.py --out /tmp/pwv/$name.inc "$@" || exit 1
  fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -Wno-unused-value -w \
     -DSAVAD_PW_INC="\"/tmp/pwv/$name.inc\"" voice_activity_detection_amd/csrc/savad.hip -o scripts/ubench/libsavad_pw_$name.so &
done
wait
ls -la scripts/ubench/libsavad_pw_*.so

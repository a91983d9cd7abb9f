#!/bin/bash
# stage time of the persistent bf16 input kernel with parts removed (results WRONG by design): scripts/ubench/abl/libsavad_a<N>.so,
# N = 16 no q/k/v^T stores, 32 no residual store, 64 no feature loads, 112 all three
for a in "$@"; do
  echo "== ablate $a"
  SAVAD_LIB=$PWD/scripts/ubench/abl/libsavad_a$a.so python scripts/ubench/input_p_ab.py --one 256 800 2>&1 | grep INPUT_P
done

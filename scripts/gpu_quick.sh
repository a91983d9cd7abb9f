#!/bin/bash
# usage: gpu_quick.sh <tag> [pytest -k expr]   -- GPU tests + the bench line (with its secondary legs)
set -u
TAG=${1:-q}; KEXPR=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$KEXPR" ]; then
  ( timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -15 ) > $OUT/pytest.log
else
  ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
fi
( timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" ) > $OUT/bench.json
tail -4 $OUT/pytest.log
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print("fp32 [32,800]: ms", d["ms_per_step"], "min", d["ms_per_step_min"], d["roofline"]["kernels_ms"])
for k,v in d.get("secondary",{}).items(): print(k, v.get("ms_per_step", v.get("ms_per_clip")), v.get("error"), (v.get("roofline") or {}).get("kernels_ms"))
PY

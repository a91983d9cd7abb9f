#!/bin/bash
# round 4, GPU call 3: the whole GPU suite on the committed sources, bench.py (N = 1, and through RCCL with one rank), host
# overhead of the module call, half batches in flight, memory-side counters that exist on this box
set -u
OUT=gpurun_out/r4c3; mkdir -p $OUT
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_all.log; tail -6 $OUT/pytest_all.log
timeout 200 python scripts/ubench/host_overhead.py > $OUT/host_overhead.log 2>&1; cat $OUT/host_overhead.log
timeout 300 python scripts/ubench/half_batches.py > $OUT/half_batches.log 2>&1; cat $OUT/half_batches.log
( timeout 600 python bench.py 2>&1 | grep "^{" ) > $OUT/bench.json; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print("fp32 [32,800]: value", d["value"], "ms", d["ms_per_step"], "one fwd", d["ms_one_forward"], "in_flight", d["in_flight"], d["roofline"]["kernels_ms"])
for k,v in d.get("secondary",{}).items(): print(k, v.get("ms_per_step", v.get("ms_per_clip", v.get("ms_per_pass"))), v.get("ms_per_step_one_in_flight"), v.get("error"), (v.get("roofline") or {}).get("kernels_ms"))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
( SAVAD_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --no-cpu-baseline 2>&1 | grep "^{" ) > $OUT/bench_dist1.json; python - <<PY
import json
d=json.loads(open("$OUT/bench_dist1.json").read())
print("RCCL world 1: value", d["value"], "ms", d["ms_per_step"], "in_flight", d["in_flight"], {k:v for k,v in d.items() if k.startswith("gather_")}, d["collective_counts"], "config3", d["config3"]["ms_per_step"], d["config3"]["finite"], d["finite"])
PY
cd /tmp; rocprofv3 --list-avail 2>/dev/null | grep -iE "MALL|HBM|DRAM|UMC|EA_RD|EA_WR|TCC_EA|MEM_" | head -80 > $OUT/counters.txt; cd - >/dev/null; wc -l $OUT/counters.txt; head -50 $OUT/counters.txt

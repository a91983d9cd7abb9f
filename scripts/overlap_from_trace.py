#!/usr/bin/env python3
"""How much do the kernels of concurrent forwards overlap?  Reads a rocprofv3 --kernel-trace CSV of a bench.py run and prints, for
the library's kernels: dispatches, average duration, the busy time (union of the kernel intervals), the summed durations, and their
ratio = average number of kernels running at once while anything runs.  usage: overlap_from_trace.py <..._kernel_trace.csv> [label]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name") or r.get("kernel_name") or ""
        if "savad" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].replace("void ", "").replace("savad::", ""), r.get("Queue_Id") or r.get("Stream_Id") or "?"))
rows.sort()
label = sys.argv[2] if len(sys.argv) > 2 else ""
if not rows:
    sys.exit("no savad kernels in the trace")
# the steady part: drop the first 20 % (warm-up, tuning) -- the rest is dominated by the timed blocks
cut = rows[len(rows) // 5][0]
rows = [r for r in rows if r[0] >= cut]
per = defaultdict(list)
for s, e, n, q in rows:
    per[n].append(e - s)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _, _ in rows)
print(f"# kernel overlap, {label}: {len(rows)} dispatches on {len(set(q for *_, q in rows))} queues over {1e-6 * (rows[-1][1] - rows[0][0]):.1f} ms")
print(f"busy time (union of kernel intervals) {busy * 1e-6:.2f} ms, summed kernel durations {total * 1e-6:.2f} ms -> {total / busy:.2f} kernels running at once on average")
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:44s} dispatches {len(v):6d}  avg {sum(v) / len(v) * 1e-3:8.1f} us  min {min(v) * 1e-3:8.1f} us")

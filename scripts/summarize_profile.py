#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel trace stats + PMC passes) into a short text summary."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    m = re.search(r"savad::(?:bf::|fs::|mel::)?(\w+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    # rocprofv3 leaves some template instantiations mangled: _ZN5savad2bf15row_kernel_bf16ILb0ELi4EEEv...
    m = re.match(r"_ZN5savad(?:2bf|2fs)?\d+([A-Za-z_0-9]+?)I((?:Lb[01]E|Li\d+E|DF16b|f)+)E", name)
    if m:
        args = []
        for a in re.findall(r"Lb[01]E|Li\d+E|DF16b|f", m.group(2)):
            args.append({"Lb0E": "false", "Lb1E": "true", "DF16b": "__bf16", "f": "float"}.get(a, a[2:-1]))
        return m.group(1) + "<" + ", ".join(args) + ">"
    return name[:60]


print(f"# rocprofv3 summary for {out}")
for f in find("trace/**/*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, out)})")
    with open(f) as fh:
        for i, row in enumerate(csv.DictReader(fh)):
            if i >= 18:
                break
            print(f"{short(row['Name']):28s} calls={row['Calls']:>5s} total_ns={row['TotalDurationNs']:>12s} "
                  f"avg_ns={float(row['AverageNs']):>12.0f} min={row['MinNs']:>9s} max={row['MaxNs']:>9s} pct={row['Percentage']}")

for f in find("trace/**/*kernel_trace.csv"):
    # per-kernel resource columns
    seen = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            if k not in seen:
                seen[k] = row
    print("\n## dispatch resources")
    for k, row in seen.items():
        cols = {c: row.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                         "Workgroup_Size", "Grid_Size")}
        print(f"{k:28s} {cols}")

for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_l2", "pmc_inst", "pmc_ea"):
    files = find(f"{sub}/**/*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    print(f"\n## {sub}: per-dispatch averages")
    for k in acc:
        vals = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
        print(f"{k:28s} " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(vals.items())))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals and vals["SQ_BUSY_CYCLES"] > 0:
            # MFMA busy is summed over SIMDs(4/CU x 256), SQ_BUSY_CYCLES over XCD-level SQs: report raw ratio + per-GUI ratio
            if "GRBM_GUI_ACTIVE" in vals and vals["GRBM_GUI_ACTIVE"] > 0:
                # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; MFMA busy is summed over all 1024 SIMDs
                cyc = vals["GRBM_GUI_ACTIVE"] / 8
                print(f"{'':28s} cycles/XCD = {cyc:.4g}; MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs) = "
                      f"{vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}")


# machine-readable per-kernel average durations of the trace pass (consumed by bench.py: event_over_rocprof)
import json
kavg = {}
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "savad" in row["Name"]:
                kavg[short(row["Name"])] = float(row["AverageNs"])
if kavg:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from bench import kernel_source_hash
        with open(os.path.join(out, "kernel_avg.json"), "w") as fh:
            json.dump({"csrc_hash": kernel_source_hash(), "kernels": kavg}, fh, indent=1, sort_keys=True)
    except Exception as exc:  # noqa: BLE001
        print("could not write kernel_avg.json:", exc)

# machine-readable HBM-side traffic per launch (consumed by bench.py's roofline.traffic)
traffic = {}
for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(f"{sub}/**/*counter_collection.csv"):
        acc, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == cname and "savad" in row["Kernel_Name"]:
                    acc[short(row["Kernel_Name"])] += float(row["Counter_Value"])
                    cnt[short(row["Kernel_Name"])] += 1
        for k in acc:
            traffic.setdefault(k, {})[cname + "_KB_per_launch"] = acc[k] / cnt[k]
for k, v in list(traffic.items()):
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 -> doubled;
    # WRITE_SIZE is taken as reported (uncalibrated).  Units: KB.
    v["hbm_bytes_per_launch"] = (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0
if traffic:
    # stamp the kernel sources the counters were measured on: bench.py quotes the traffic only for matching code
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from bench import kernel_source_hash
        traffic["csrc_hash"] = kernel_source_hash()
    except Exception as exc:  # noqa: BLE001
        print("could not stamp csrc_hash:", exc)
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1, sort_keys=True)

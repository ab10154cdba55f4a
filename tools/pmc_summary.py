#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile_gpu.sh: per-kernel duration stats and
per-dispatch mean of every PMC counter, for the rtamd kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print(f"# rocprofv3 summary for {root}")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel-trace --stats (rtamd kernels)")
    for row in csv.DictReader(open(f)):
        if "rtamd" in row["Name"]:
            print(f'{row["Name"][:70]:70s} calls={row["Calls"]:>5s} avg_ns={float(row["AverageNs"]):>12.1f} '
                  f'min_ns={row["MinNs"]:>9s} max_ns={row["MaxNs"]:>9s} pct={row["Percentage"]}')
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rtamd" in k and k not in seen:
            seen.add(k)
            print(f'{k[:70]:70s} VGPR={row["VGPR_Count"]} AGPR={row["Accum_VGPR_Count"]} SGPR={row["SGPR_Count"]} '
                  f'LDS={row["LDS_Block_Size"]} scratch={row["Scratch_Size"]} wg={row["Workgroup_Size_X"]} grid={row["Grid_Size_X"]}')
print("## PMC (mean per dispatch of the dominant kernel; separate passes)")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        names = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "rtamd" not in k or "<0, true>" in k or "generate_rays" in k:
                continue
            per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = k
        for did, cs in per_dispatch.items():
            for c, v in cs.items():
                acc[names[did]][c].append(v)
    for k, cs in acc.items():
        print(f"[{os.path.basename(d)}] {k[:60]}")
        for c, vs in sorted(cs.items()):
            print(f"    {c:36s} {sum(vs) / len(vs):18.1f}   (n={len(vs)})")

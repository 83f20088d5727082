"""Aggregates a rocprofv3 --kernel-trace CSV by (kernel, grid size): count, average / min duration in us -- launches of one kernel on
different layer shapes have different grids.  python tools/kernel_by_grid.py <kernel_trace.csv> [name filter]"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = name.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        rows[(short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("LDS_Block_Size", ""))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':48s} {'workgroups':>10s} {'lds':>7s} {'n':>5s} {'avg us':>9s} {'min us':>9s} {'total ms':>9s}")
for (k, g, lds), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:48]:48s} {g:10d} {lds:>7s} {len(v):5d} {sum(v) / len(v):9.1f} {min(v):9.1f} {sum(v) / 1e3:9.2f}")

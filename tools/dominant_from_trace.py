#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: launch statistics of the dominant kernel (the fused LK
iteration, k_lkw<HW, MODE_ITER>, at the finest level = its largest grid), which the per-name
--stats summary folds together with the coarser levels' launches.
Usage: python3 tools/dominant_from_trace.py <dir with *_kernel_trace.csv> <out.json>"""
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
match = sys.argv[3] if len(sys.argv) > 3 else "k_lkw<"      # e.g. "k_lks<" for the streaming kernel of the tolerant mode
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"] and (", 1, " in r["Kernel_Name"] or "<1, " in r["Kernel_Name"]):
            grid = int(r.get("Grid_Size") or 0) or int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            rows.append((r["Kernel_Name"], grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
big = max(g for _, g, _ in rows)
d = [t for _, g, t in rows if g == big]
res = {"kernel": rows[0][0], "selection": "launches with the largest grid (finest pyramid level)", "grid_threads": big,
       "launches": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d),
       "all_levels_launches": len(rows), "source": "rocprofv3 --kernel-trace --stats (same command as the stats CSV)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))

#!/usr/bin/env python3
"""Per dispatch: vector-ALU occupancy from the SQ counters (development tool; tools/pmc_calib.sh).
busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * launch cycles); cycles per instruction = SQ_ACTIVE_INST_VALU * 4 / SQ_INSTS_VALU."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
for sub in ("ubench", "lk"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-46:], int(r["Grid_Size"]))
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"== {sub}")
    agg = collections.defaultdict(list)
    for (did, name, grid), c in sorted(rows.items()):
        if "SQ_INSTS_VALU" not in c or c["SQ_INSTS_VALU"] < 1e5:
            continue
        clk = c["GRBM_GUI_ACTIVE"] / 8.0 / c["_us"] / 1e3   # GHz (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        busy = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
        cpi = c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"]
        wait = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        stall = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        agg[(name, grid)].append((c["_us"], clk, busy, cpi, wait, stall, c["SQ_INSTS_VALU"]))
    for (name, grid), v in agg.items():
        n = len(v)
        m = [sum(x[i] for x in v) / n for i in range(7)]
        print(f"{name:48s} grid {grid:9d} x{n:3d}  {m[0]:9.1f} us  clock {m[1]:.2f} GHz  VALU busy {m[2]*100:6.1f} %  "
              f"cycles/VALU-instr {m[3]:5.2f}  wave-time waiting {m[4]*100:5.1f} %  issue-stalled {m[5]*100:5.1f} %  VALU instr {m[6]:.4g}")

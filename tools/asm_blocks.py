#!/usr/bin/env python3
"""Static instruction mix per basic block of one kernel in oflk_gfx950.s (development tool).
Usage: python3 tools/asm_blocks.py <mangled-name-prefix> [asm file]"""
import collections
import sys

name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "optical-flow-fpga_amd/csrc/oflk_gfx950.s"
lines = open(path).read().split("\n")
st = [i for i, l in enumerate(lines) if l.startswith(name) and ":" in l][0]
en = [i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
seg = 0
cnt = collections.defaultdict(collections.Counter)
for l in lines[st + 1:en]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".p2align"):
        continue
    op = l.split()[0]
    if op.endswith(":"):
        seg += 1
        cnt[seg]["LABEL " + op] += 1
        continue
    if op == "s_barrier":
        seg += 1
        cnt[seg]["BARRIER"] += 1
        continue
    if op.startswith("s_cbranch") or op == "s_branch":
        cnt[seg]["BR " + l] += 1
    cnt[seg][op] += 1
tot = collections.Counter()
for sgi in sorted(cnt):
    c = cnt[sgi]
    v = sum(n for o, n in c.items() if o.startswith("v_"))
    f64 = sum(n for o, n in c.items() if o.startswith("v_") and "f64" in o)
    sal = sum(n for o, n in c.items() if o.startswith("s_"))
    ds = sum(n for o, n in c.items() if o.startswith("ds_"))
    gl = sum(n for o, n in c.items() if o.startswith("global_") or o.startswith("buffer_"))
    tags = [o for o in c if o.startswith("LABEL") or o.startswith("BR") or o == "BARRIER"]
    print(f"seg {sgi}: valu {v} (f64 {f64}) salu {sal} ds {ds} vmem {gl}  {tags}")
    if "-v" in sys.argv:
        print("      " + ", ".join(f"{o}:{n}" for o, n in c.most_common(40) if o.startswith("v_")))

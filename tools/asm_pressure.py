#!/usr/bin/env python3
"""Approximate VGPR pressure along one kernel's assembly (development tool): treats the code as
straight-line, a register live from each write to its last read before the next write.
Usage: python3 tools/asm_pressure.py <mangled-name-prefix> [asm file]"""
import re
import sys

name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "optical-flow-fpga_amd/csrc/oflk_gfx950.s"
lines = open(path).read().split("\n")
st = [i for i, l in enumerate(lines) if l.startswith(name) and ":" in l][0]
en = [i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
ins = []
for i in range(st + 1, en):
    l = lines[i].split(";")[0].strip()
    if not l or l.startswith("."):
        if l.endswith(":"):
            ins.append((i, l, [], []))
        continue
    m = re.match(r"(\S+)\s*(.*)", l)
    op, rest = m.group(1), m.group(2)
    ops = [o.strip() for o in rest.split(",")] if rest else []
    def regs(o):
        r = []
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", o):
            r += list(range(int(a), int(b) + 1))
        r += [int(x) for x in re.findall(r"\bv(\d+)\b", o)]
        return r
    is_store = op.startswith("global_store") or op.startswith("ds_write") or op.startswith("buffer_store") or op.startswith("v_cmp") or op.startswith("s_") or op.startswith("ds_write")
    if is_store or not ops:
        d, u = [], [x for o in ops for x in regs(o)]
    else:
        d, u = regs(ops[0]), [x for o in ops[1:] for x in regs(o)]
        if op.startswith("v_fmac") or op.startswith("v_mac") or "accum" in op:
            u += d
    ins.append((i, l, d, u))
n = len(ins)
live_until = {}
live = [0] * n
# backward pass: live set
cur = set()
for k in range(n - 1, -1, -1):
    _, _, d, u = ins[k]
    for r in d:
        cur.discard(r)
    for r in u:
        cur.add(r)
    live[k] = len(cur)
mx = max(live)
print("max live (approx):", mx)
step = max(1, n // 60)
for k in range(0, n, step):
    seg = live[k:k + step]
    j = k + seg.index(max(seg))
    print(f"{ins[j][0]:7d} live={max(seg):4d}  {ins[j][1][:70]}")

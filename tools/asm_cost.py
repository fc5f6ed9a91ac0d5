#!/usr/bin/env python3
"""Static VALU cost of one kernel's basic blocks priced with the wall-clock-validated gfx950 table
(tools/ubench/valu_wall.hip, profiles/r03_valu_wall.txt): SIMD cycles per wave64 instruction.
Usage: python3 tools/asm_cost.py <mangled-name-prefix> [asm file] [-v]"""
import collections
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32",
        "v_max_f32", "v_min_f32"}
C_FAST, C_SLOW, C_RCP = 2.4, 4.15, 8.1


def cost(op):
    base = op
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if base.endswith(suf):
            base = base[: -len(suf)]
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return C_SLOW
    if base in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32"):
        return C_RCP
    if base in FAST:
        return C_FAST
    return C_SLOW


def main():
    name = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else "optical-flow-fpga_amd/csrc/oflk_gfx950.s"
    lines = open(path).read().split("\n")
    st = [i for i, l in enumerate(lines) if l.startswith(name) and ":" in l][0]
    en = [i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    seg = 0
    cnt = collections.defaultdict(collections.Counter)
    tags = collections.defaultdict(list)
    for l in lines[st + 1:en]:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".p2align"):
            continue
        op = l.split()[0]
        if op.endswith(":"):
            seg += 1
            tags[seg].append(op)
            continue
        if op == "s_barrier":
            seg += 1
            tags[seg].append("BARRIER")
            continue
        cnt[seg][op] += 1
    for s in sorted(cnt):
        c = cnt[s]
        v = sum(n for o, n in c.items() if o.startswith("v_"))
        cyc = sum(n * cost(o) for o, n in c.items() if o.startswith("v_"))
        slow = sum(n for o, n in c.items() if o.startswith("v_") and cost(o) > C_FAST)
        ds = sum(n for o, n in c.items() if o.startswith("ds_"))
        vm = sum(n for o, n in c.items() if o.startswith("global_") or o.startswith("buffer_") or o.startswith("flat_"))
        sa = sum(n for o, n in c.items() if o.startswith("s_"))
        if v + ds + vm < 3:
            continue
        print(f"seg {s:3d}: valu {v:4d} ({slow:4d} slow)  cycles {cyc:7.0f}   salu {sa:3d} ds {ds:3d} vmem {vm:3d}  {' '.join(tags[s])}")
        if "-v" in sys.argv:
            print("        " + ", ".join(f"{o}:{n}" for o, n in c.most_common(30) if o.startswith("v_")))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Instruction-issue bounds of the dominant kernel (finest-level fused LK iteration), development tool.

The kernel is a stencil with exact (NumPy/SciPy-order) arithmetic: besides the HBM bound there are two
bounds on the instruction side, both derived from measurements:

  valu_pipe    the vector ALU of a SIMD is occupied  sum_i n_i * c_i  cycles per wave and tile, with
               n_i the wave-instructions of class i (dynamic SQ_INSTS_VALU of the launch, split by the
               static opcode mix of the kernel's ISA) and c_i the saturated SIMD cycles per
               wave-instruction of that class (tools/ubench/valu_cycles.hip, column w8)
  issue_cadence  one wave issues at most one instruction per ~5 cycles (same microbenchmark, column w1:
               5.0-5.6 cycles for every VALU class), so a SIMD holding W waves retires at most W/5
               instructions per cycle; all instruction types of the launch count

Inputs: a directory written by tools/pmc_sq.sh (rocprofv3 --pmc passes over tools/kbench.py), the ISA
from `make -C optical-flow-fpga_amd/csrc asm`, and profiles/<tag>_valu_cycles.txt.
Usage: python3 tools/issue_bounds.py gpurun_out/pmc_<tag> profiles/<tag>_valu_cycles.txt profiles/<tag>_issue_bounds.json
"""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
KERNEL = "_ZN4oflk5k_lkwILi2ELi1ELb1EfEEvNS_6LkArgsE"   # k_lkw<2, MODE_ITER, true, float>
WAVES_PER_SIMD = 4
N_SIMD = 1024
CADENCE = 5.0


def cycle_table(path):
    t = {}
    for line in open(path):
        m = re.match(r"(\S+(?: \S+)*?)\s+w1:\s*([\d.]+)\s+w2:\s*([\d.]+)\s+w4:\s*([\d.]+)\s+w8:\s*([\d.]+)", line)
        if m:
            t[m.group(1)] = {"w1": float(m.group(2)), "w4": float(m.group(4)), "w8": float(m.group(5))}
    return t


def op_class(op, tbl):
    """saturated SIMD cycles per wave-instruction of an opcode, from the measured table"""
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    w8 = lambda k: tbl[k]["w8"]
    if "f64" in base and base.startswith("v_cvt"):
        return w8("v_cvt_f64_f32")
    if base == "v_floor_f64":
        return w8("v_floor_f64")
    if base.endswith("_u64") or base.endswith("_i64"):
        return w8("v_cmp_le_u64") if base.startswith("v_cmp") else w8("v_add_f64")
    if "f64" in base or base.startswith("v_pk_") and "f32" in base or base == "v_mov_b64":
        return w8("v_add_f64")
    if base == "v_rcp_f32":
        return w8("v_rcp_f32")
    if base == "v_div_scale_f32":
        return w8("v_div_scale_f32")
    if base in ("v_div_fmas_f32", "v_div_fixup_f32"):
        return w8("v_div_fmas_f32")
    if op.endswith("_dpp"):
        return w8("v_mov_b32_dpp")
    if base in ("v_fma_f32", "v_fmac_f32"):
        return w8("v_fma_f32")
    if base in ("v_mul_u32_u24", "v_mul_i32_i24", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3"):
        return w8("v_mul_u32_u24")
    if base in ("v_mad_u32_u24", "v_mad_i32_i24", "v_med3_i32", "v_lshl_add_u32", "v_add_lshl_u32", "v_add3_u32", "v_lshl_or_b32",
                "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_cndmask_b32", "v_bfe_u32", "v_and_or_b32",
                "v_xad_u32", "v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32") or base.startswith("v_cmp"):
        return w8("v_mad_u32_u24")
    return w8("v_add_f32")   # plain VOP1/VOP2: add, mul, sub, mov, shifts, min/max, logic


def static_mix(asm_path, tbl):
    lines = open(asm_path).read().split("\n")
    st = [i for i, l in enumerate(lines) if l.startswith(KERNEL + ":")][0]
    en = [i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    cnt = collections.Counter()
    for l in lines[st + 1:en]:
        l = l.strip()
        if l.startswith("v_"):
            cnt[l.split()[0]] += 1
    tot = sum(cnt.values())
    mean_cost = sum(n * op_class(op, tbl) for op, n in cnt.items()) / tot
    return tot, mean_cost, cnt


def main():
    pmc_dir, table_path, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    tbl = cycle_table(table_path)
    asm = ROOT / "optical-flow-fpga_amd" / "csrc" / "oflk_gfx950.s"
    n_static, mean_cost, cnt = static_mix(asm, tbl)
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_lkw<2, 1, true" not in r["Kernel_Name"]:
                continue
            g = int(r["Grid_Size"])
            rows[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[g][(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    big = max(rows)   # finest level = largest grid
    c = {k: sum(v) / len(v) for k, v in rows[big].items()}
    us = sum(dur[big].values()) / len(dur[big])
    clock_ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)          # sum over 8 XCDs / wall
    valu = c["SQ_INSTS_VALU"]
    total = valu + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
    pipe_cycles = valu * mean_cost / N_SIMD
    cadence_cycles = total * CADENCE / (N_SIMD * WAVES_PER_SIMD)
    res = {
        "kernel": "k_lkw<2, MODE_ITER, true> finest level, 32 x 1920x1080 (tools/kbench.py under rocprofv3 --pmc)",
        "pairs": 32, "shape": [1080, 1920],
        "launch_us_under_pmc": round(us, 1), "clock_GHz": round(clock_ghz, 3),
        "wave_instructions_per_launch": {"valu": valu, "salu": c.get("SQ_INSTS_SALU"), "lds": c.get("SQ_INSTS_LDS"),
                                         "vmem_rd": c.get("SQ_INSTS_VMEM_RD"), "vmem_wr": c.get("SQ_INSTS_VMEM_WR"), "all": total},
        "static_valu_instructions": n_static, "mean_saturated_cycles_per_valu_instruction": round(mean_cost, 3),
        "valu_pipe": {"floor_us": round(pipe_cycles / clock_ghz / 1e3, 1), "frac": round(pipe_cycles / clock_ghz / 1e3 / us, 3),
                      "meaning": "vector-ALU occupancy of the launch's instruction mix at saturated per-instruction rates"},
        "issue_cadence": {"floor_us": round(cadence_cycles / clock_ghz / 1e3, 1),
                          "frac": round(cadence_cycles / clock_ghz / 1e3 / us, 3), "waves_per_simd": WAVES_PER_SIMD,
                          "cycles_per_instruction_per_wave": CADENCE,
                          "meaning": "one wave issues <= 1 instruction per ~5 cycles; 4 resident waves per SIMD (127 VGPRs, 38 KB LDS per block)"},
        "wave_state_shares": {k: round(c[k] / c["SQ_WAVE_CYCLES"], 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")
                              if k in c and "SQ_WAVE_CYCLES" in c},
        "sources": [str(Path(pmc_dir).name), str(Path(table_path).name), "oflk_gfx950.s (make asm)"],
    }
    Path(out_path).write_text(json.dumps(res, indent=1))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

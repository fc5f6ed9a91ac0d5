#!/usr/bin/env python3
"""Instruction-issue bounds of the dominant kernel (finest-level fused LK iteration), development tool.

The kernel is a stencil with exact (NumPy/SciPy-order) arithmetic: besides the HBM bound there are two
bounds on the instruction side, both derived from measurements:

  valu_pipe    the vector ALU of a SIMD is occupied  sum_i n_i * c_i  cycles per wave and tile, with
               n_i the wave-instructions of class i (dynamic SQ_INSTS_VALU of the launch, split by the
               static opcode mix of the kernel's ISA) and c_i the saturated SIMD cycles per
               wave-instruction of that class, checked against wall-clock (tools/ubench/valu_wall.hip:
               2.3 - 2.5 for plain fp32 / integer add / logic / move (the lowest member is the price), 4.15 for everything else,
               8.1 for v_rcp_f32;
               round 2's table, from per-wave s_memtime deltas over an assumed occupancy, was 2.5x too low)
  issue_cadence  one wave issues at most one instruction per ~5 cycles (same microbenchmark, column w1:
               5.0-5.6 cycles for every VALU class), so a SIMD holding W waves retires at most W/5
               instructions per cycle; all instruction types of the launch count

Inputs: a directory written by tools/pmc_sq.sh (rocprofv3 --pmc passes over tools/kbench.py), the ISA
from `make -C optical-flow-fpga_amd/csrc asm`, and profiles/<tag>_valu_wall.txt.
Usage: python3 tools/issue_bounds.py gpurun_out/pmc_<tag> profiles/<tag>_valu_wall.txt profiles/<tag>_issue_bounds.json
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
    """wall-clock-validated table (tools/ubench/valu_wall.hip): SIMD cycles per wave64 instruction at saturation (w8 rows)"""
    t = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+w8\s+wall.*?cyc/SIMD\s+([\d.]+)", line)
        if m:
            t[m.group(1).strip()] = float(m.group(2))
    return t


def price_classes(tbl):
    """three price classes: plain fp32 / integer add / logic / move; everything else; the transcendental unit"""
    # fast: the LOWEST figure among the class members (2.3 - 2.55 in one run: the census shows uneven wave placement for
    # these short kernels), so that the priced time stays a lower bound of the launch
    fast = min(tbl[k] for k in ("v_add_f32", "v_fma_f32", "v_mul_f32", "v_add_u32", "v_and_b32", "v_mov_b32") if k in tbl)
    return {"fast": fast, "slow": tbl["v_add_f64"], "rcp": tbl["v_rcp_f32"]}


def static_mix(asm_path, kernel, prices):
    sys.path.insert(0, str(ROOT / "tools"))
    import asm_cost

    lines = open(asm_path).read().split("\n")
    st = [i for i, l in enumerate(lines) if l.startswith(kernel + ":")][0]
    en = [i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    cnt = collections.Counter()
    for l in lines[st + 1:en]:
        l = l.strip()
        if l.startswith("v_"):
            cnt[l.split()[0]] += 1
    tot = sum(cnt.values())
    cls = {asm_cost.C_FAST: "fast", asm_cost.C_SLOW: "slow", asm_cost.C_RCP: "rcp"}
    by = collections.Counter()
    for op, n in cnt.items():
        by[cls[asm_cost.cost(op)]] += n
    mean_cost = sum(by[k] * prices[k] for k in by) / tot
    return tot, mean_cost, {k: by[k] / tot for k in by}


def kernel_bounds(pmc_dir, match, kernel_sym, asm, prices, hbm_bytes_per_px=None, waves_per_simd=WAVES_PER_SIMD):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    files = collections.defaultdict(list)
    for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
        files[str(Path(f).parent)].append(f)
    # one file per pass directory: the newest (a directory merged back from several runs holds their files side by side)
    for f in (max(v, key=lambda q: Path(q).stat().st_mtime) for v in files.values()):
        for r in csv.DictReader(open(f)):
            if match not in r["Kernel_Name"]:
                continue
            g = int(r["Grid_Size"])
            rows[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[g][(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    big = max(rows)   # finest level = largest grid
    c = {k: sum(v) / len(v) for k, v in rows[big].items()}
    us = sum(dur[big].values()) / len(dur[big])
    clock_ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)          # sum over 8 XCDs / wall
    n_static, mean_cost, shares = static_mix(asm, kernel_sym, prices)
    valu = c["SQ_INSTS_VALU"]
    total = valu + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
    pipe_cycles = valu * mean_cost / N_SIMD
    cadence_cycles = total * CADENCE / (N_SIMD * waves_per_simd)
    return {
        "launch_us_under_pmc": round(us, 1), "clock_GHz": round(clock_ghz, 3),
        "wave_instructions_per_launch": {"valu": valu, "salu": c.get("SQ_INSTS_SALU"), "lds": c.get("SQ_INSTS_LDS"),
                                         "vmem_rd": c.get("SQ_INSTS_VMEM_RD"), "vmem_wr": c.get("SQ_INSTS_VMEM_WR"), "all": total},
        "static_valu_instructions": n_static, "static_valu_class_shares": {k: round(v, 3) for k, v in shares.items()},
        "mean_saturated_cycles_per_valu_instruction": round(mean_cost, 3),
        "valu_pipe": {"floor_us": round(pipe_cycles / clock_ghz / 1e3, 1), "frac": round(pipe_cycles / clock_ghz / 1e3 / us, 3),
                      "meaning": "vector-ALU time of the launch's instructions at the saturated, wall-clock-validated rate of their class"},
        "issue_cadence": {"floor_us": round(cadence_cycles / clock_ghz / 1e3, 1),
                          "frac": round(cadence_cycles / clock_ghz / 1e3 / us, 3), "waves_per_simd": waves_per_simd,
                          "cycles_per_instruction_per_wave": CADENCE,
                          "meaning": f"one wave issues <= 1 instruction per ~5 cycles; {waves_per_simd} resident waves per SIMD"},
        "wave_state_shares": {k: round(c[k] / c["SQ_WAVE_CYCLES"], 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")
                              if k in c and "SQ_WAVE_CYCLES" in c},
    }


def main():
    pmc_dir, table_path, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    prices = price_classes(cycle_table(table_path))
    asm = ROOT / "optical-flow-fpga_amd" / "csrc" / "oflk_gfx950.s"
    res = kernel_bounds(pmc_dir, "k_lkw<2, 1, true", KERNEL, asm, prices)
    res = dict({"kernel": "k_lkw<2, MODE_ITER, true> finest level, 32 x 1920x1080 (tools/kbench.py under rocprofv3 --pmc)",
                "pairs": 32, "shape": [1080, 1920], "price_classes_cycles": prices}, **res)
    try:
        res["pyr_down"] = kernel_bounds(pmc_dir, "k_pyr_down<float", "_ZN4oflk10k_pyr_downIfLb0EEEvNS_7PyrArgsE", asm, prices)
    except Exception as e:   # counters of that kernel missing
        res["pyr_down"] = {"error": str(e)}
    # round 4: the single-scale kernels on their own (32 pairs of 1080p, tools/profiles_r04.sh) and the streaming iteration
    # kernel of the tolerant mode; HBM floor of a launch = algorithmic bytes / 8 TB/s beside the instruction-side floors
    if len(sys.argv) > 4:
        single_dir = sys.argv[4]
        px = 32 * 1080 * 1920
        for key, sub, match, sym, waves, bpp in (
                ("single_tile_5x5", "tile5", "k_lkw<2, 0, true", "_ZN4oflk5k_lkwILi2ELi0ELb1EfEEvNS_6LkArgsE", 4, 16),
                ("single_tile_7x7", "tile7", "k_lkw<3, 0, true", "_ZN4oflk5k_lkwILi3ELi0ELb1EfEEvNS_6LkArgsE", 3, 16),
                ("single_stream_5x5", "stream5", "k_lks<0, true", "_ZN4oflk5k_lksILi0ELb1ELi0EfLb0ELi2EEEvNS_6LkArgsE", 4, 16),
                ("iter_stream_5x5_tolerant", "tol", "k_lks<1, true, 1, float, false, 2", "_ZN4oflk5k_lksILi1ELb1ELi1EfLb0ELi2EEEvNS_6LkArgsE", 2, 24)):
            try:
                r = kernel_bounds(single_dir + "/" + sub, match, sym, asm, prices, waves_per_simd=waves)
                r["hbm_floor_us"] = round(px * bpp / 8e12 * 1e6, 1)
                r["hbm_frac"] = round(r["hbm_floor_us"] / r["launch_us_under_pmc"], 3)
                res[key] = r
            except Exception as e:
                res[key] = {"error": repr(e)}
    res["sources"] = [str(Path(pmc_dir).name), str(Path(table_path).name), "oflk_gfx950.s (make asm)", "tools/asm_cost.py (opcode classes)"]
    Path(out_path).write_text(json.dumps(res, indent=1))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

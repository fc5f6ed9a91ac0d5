#!/usr/bin/env python3
"""Index of profiles/<tag>_*: what each file is and the headline numbers it carries (development tool).
Usage: python3 tools/profiles_summary.py r03   -> profiles/r03_SUMMARY.md"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    P = ROOT / "profiles"
    b = json.loads((P / f"{tag}_bench.json").read_text())
    r = b["roofline"]
    c = json.loads((P / f"{tag}_configs.json").read_text())
    k = json.loads((P / f"{tag}_rocprofv3_dominant_kernel.json").read_text())
    bt = json.loads((P / f"{tag}_block_times.json").read_text())
    rows = "\n".join(f"| {x['config']} | {x['mode']} | {x['pairs']} | {x['us_per_call']} | {x['Mpix_per_s']} | {x.get('frac_of_8TBs', '')} |" for x in c["rows"])
    txt = f"""# profiles/{tag} — what each file is and the numbers it carries (one MI355X box of the pool; boxes differ by 2-10 %)

| file | made by | content |
|---|---|---|
| `{tag}_bench.json` | `python bench.py` (the driver's command) | the contract line: **{b['value']} Mpix/s**, {b['config']['pairs_per_gpu_per_step']} pairs of 1080p per step, {b['ms_per_step']} ms/step; `roofline`: dominant kernel {r['avg_launch_us']} µs per launch, {r['achieved']} GB/s algorithmic = **{r['frac']}** of 8 TB/s, HBM traffic {r['traffic'] / 1e9:.2f} GB per launch ({'measured during the run by rocprofv3 --pmc child passes' if str(r['traffic_source']).startswith('measured') else 'from ' + str(r['traffic_source'])[:40]}) = {r['traffic'] / r['algorithmic_bytes_per_launch']:.2f}x algorithmic; `valu_pipe` {r['valu_pipe']['frac']}, `issue_cadence` {r['issue_cadence']['frac']}; `roofline_pyr` {b['roofline_pyr']['frac']}; `cpu_baseline` {b['cpu_baseline']['value']} Mpix/s (oracle, 1 thread) |
| `{tag}_bench_under_rocprof.json`, `{tag}_rocprofv3_kernel_stats.csv`, `{tag}_rocprofv3_dominant_kernel.json` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-one-pair --no-live-traffic` | per-kernel statistics of the bench command; finest-level launches of the dominant kernel: {k['avg_ns'] / 1e3:.0f} µs average over {k['launches']} launches (agrees with the HIP-event figure of the bench line) |
| `{tag}_hbm_traffic.json`, `{tag}_pmc/` | `tools/measure_traffic.sh` | FETCH_SIZE / WRITE_SIZE of the dominant kernel in separate `--pmc` passes over the bench command (fallback of `roofline.traffic`) |
| `{tag}_sq_counters.txt`, `{tag}_issue_bounds.json`, `{tag}_valu_wall.txt`, `{tag}_sq_counter_calibration.txt` | `tools/pmc_sq.sh`, `tools/issue_bounds.py`, `tools/ubench/valu_wall`, `tools/pmc_calib.sh` | SQ instruction / wait counters of every LK kernel (32-pair launches); the instruction-side bounds of the iteration kernel and of `k_pyr_down`; SIMD cycles per VALU instruction form checked against wall-clock, with a census of where the waves ran; what the SQ VALU-busy counter means (one quad-cycle per instruction, whatever the instruction) |
| `{tag}_corun.json` | `tools/corun.py` | the step beside a partner kernel that holds wave slots / issues v_add_f32 / v_add_f64 / copies HBM on a second stream |
| `{tag}_contracted_epe.json`, `{tag}_host_latency.txt` | `tests/test_gpu_round3.py`, `tools/host_latency.py` | opt-in contracted arithmetic: flow values that differ from the exact result, pyramid kernel time; host-to-host timings (single pairs, and 32 pairs in one chunked call) |
| `{tag}_stamps_timeline.{{txt,json}}`, `{tag}_block_times.json` | `tools/stamps.py`, `tools/block_times.py` (diagnostic builds) | where a wave spends a tile's time (13 sections); block lifetimes: slot occupancy {bt['slot_occupancy_mean']}, {bt['per_block_us']['per_tile']} µs per tile, prologue {bt['per_block_us']['prologue']} µs, refill gap {bt['slot_refill_gap_us']['mean']} µs |
| `{tag}_one_pair_timeline.txt` | `rocprofv3 --kernel-trace -- python3 tools/kbench.py --pairs 1` | the 14 dependent launches of ONE 1080p pair per call, and the call with 64 x 8 / 16 / 24 tiles |
| `{tag}_grid_barrier.txt` | `tools/ubench/grid_barrier` | a dependent kernel launch (2.3 - 2.9 µs) against a device-scope grid barrier inside one launch (8 - 107 µs for 96 - 1 024 blocks) |
| `{tag}_configs.json` | `tools/measure_configs.py` | every BASELINE config that fits one GPU + the f3 integer mode (table below) |
| `{tag}_bench_4k64_1gpu.json` | `python bench.py --config 4k64` | BASELINE configs[3] (64 pairs of 4K, one job) on one GPU |
| `{tag}_fp16_epe.json`, `{tag}_pmc_fp16.txt`, `{tag}_rowwalk.txt`, `{tag}_hbm_probe.txt` | `tests/test_gpu_fp16.py`, `tools/pmc_fp16.sh`, `tools/ubench/rowwalk`, `tools/hbm_probe.py` | fp16 mode: EPE against the exact path per pattern; counters of the streaming kernel; what the HBM gives the same read/write mix by access pattern; library elementwise kernels as the practical ceiling |

| config | mode | pairs | µs per call | Mpix/s | fraction of 8 TB/s (single-scale rows) |
|---|---|---|---|---|---|
{rows}
"""
    (P / f"{tag}_SUMMARY.md").write_text(txt)


if __name__ == "__main__":
    main()

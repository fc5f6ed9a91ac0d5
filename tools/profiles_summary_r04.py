#!/usr/bin/env python3
"""profiles/r04_SUMMARY.md: what each round-4 profile file is and the headline numbers it carries (development tool)."""
import json
from pathlib import Path

P = Path(__file__).resolve().parents[1] / "profiles"
b = json.load(open(P / "r04_bench.json"))
t = b["tolerance_mode"]
c = json.load(open(P / "r04_configs.json"))
k = json.load(open(P / "r04_rocprofv3_dominant_kernel.json"))
kt = json.load(open(P / "r04_rocprofv3_tolerant_kernel.json"))
tr = json.load(open(P / "r04_hbm_traffic.json"))
trt = json.load(open(P / "r04_hbm_traffic_tolerant.json"))
ss = b.get("single_scale_batched") or {}
rows = "\n".join(f"| {x['config']} | {x['mode']} | {x['pairs']} | {x['us_per_call']} | {x['Mpix_per_s']} | {x.get('frac_of_8TBs', '')} |" for x in c["rows"])
txt = f"""# profiles/r04 -- what each file is and the numbers it carries (one MI355X box of the pool; boxes differ by 2-10 %)

Made by `bash tools/profiles_r04.sh` on the GPU box (the ablation files by `tools/experiments/fast_mode_ablation.py` on the CPU).
The round-3 evidence the exact path's analysis rests on (cycle table against wall-clock, co-run, stamps, block times, grid
barrier) stays under `profiles/r03_*`: the exact kernels did not change this round.

| file | made by | content |
|---|---|---|
| `r04_bench.json` | `python bench.py` | the contract line: **{b['value']} Mpix/s** exact ({b['ms_per_step']} ms per 128-pair step; dominant kernel {b['roofline']['avg_launch_us']} us per launch = **{b['roofline']['frac']}** of 8 TB/s); `tolerance_mode`: **{t['value']} Mpix/s** ({t['ms_per_step']} ms; whole call {t['whole_call']['frac_of_peak']} of the peak), dominant kernel `k_lks` {t['roofline']['avg_launch_us']} us = **{t['roofline']['frac']}**, worst field {t['max_mean_epe_vs_reference']:.2e} px of {t['fields']} ({t['worst_field']}); `single_scale_batched`: {ss.get('Mpix/s')} Mpix/s = {ss.get('frac_of_hbm_peak')} of the peak; `contracted_arithmetic`, `one_pair_per_call`, `cpu_baseline`, `epe_vs_reference` |
| `r04_bench_under_rocprof.json`, `r04_rocprofv3_kernel_stats.csv`, `r04_rocprofv3_dominant_kernel.json`, `r04_rocprofv3_tolerant_kernel.json` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-one-pair --no-live-traffic` | per-kernel statistics of the bench command; finest-level launches: exact `k_lkw<2, ITER>` {k['avg_ns'] / 1e3:.0f} us over {k['launches']} launches, tolerant `k_lks<ITER>` {kt['avg_ns'] / 1e3:.0f} us over {kt['launches']} (agree with the HIP-event figures of the bench line) |
| `r04_hbm_traffic.json`, `r04_hbm_traffic_tolerant.json`, `r04_pmc/` | `tools/measure_traffic.sh` (FETCH_SIZE and WRITE_SIZE in separate `--pmc` passes over the bench command, FETCH_SIZE x2 on gfx950) | exact kernel: {tr['hbm_bytes_per_launch'] / 1e9:.2f} GB per 128-pair launch = {tr['hbm_bytes_per_launch'] / 6370099200:.2f}x algorithmic; streaming kernel of the tolerant mode: {trt['hbm_bytes_per_launch'] / 1e9:.2f} GB = **{trt['hbm_bytes_per_launch'] / trt['algorithmic_bytes_per_launch']:.2f}x** of 24 B/px (1.39x before the flow of an output row came back from the LDS ring; the first launch of a level reads 16 B/px) |
| `r04_sq_counters.txt`, `r04_issue_bounds.json` | `tools/pmc_sq.sh`, `tools/issue_bounds.py` (cycle table: `r03_valu_wall.txt`) | SQ counters of every LK kernel; instruction-side bounds of the exact iteration kernel and `k_pyr_down` (as round 3), and new: the single-scale kernels on their own (tile 5x5 / 7x7, streaming 5x5) and the tolerant mode's streaming iteration kernel, each with its HBM fraction beside it |
| `r04_tolerance_ablation.txt`, `.json` | `tools/experiments/fast_mode_ablation.py --frames 1080p` | mean EPE against the exact flow per relaxed stage x pyramid level x iteration, groups of cells, and the shipped combination (13 patterns + the 1080p bench pair) |
| `r04_tolerant_epe.json` | `tests/test_gpu_round4.py` | the tolerant mode's mean EPE per field against the REFERENCE's dense flows (`tests/golden/dense_reference_flows.npz`) |
| `r04_configs.json` | `tools/measure_configs.py` | every BASELINE config that fits one GPU, with the tolerant mode and the tile-kernel single-scale rows beside them (table below) |
| `r04_bench_4k64_1gpu.json` | `python bench.py --config 4k64` | BASELINE configs[3] (64 pairs of 4K, one job) on one GPU |
| `r04_host_latency.txt` | `tools/host_latency.py` | host-to-host timings (single pairs, and 32 pairs in one chunked call) |

| config | mode | pairs | us per call | Mpix/s | fraction of 8 TB/s (single-scale rows) |
|---|---|---|---|---|---|
{rows}
"""
(P / "r04_SUMMARY.md").write_text(txt)
print("written", P / "r04_SUMMARY.md")

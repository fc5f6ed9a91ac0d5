#!/usr/bin/env python3
"""bench.py -- Mpix/s of dense pyramidal Lucas-Kanade flow on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  Rank 0 prints
ONE JSON line.

Workload (BASELINE.json configs[2], the config the metric is quoted on): batches
of independent 1920x1080 frame pairs, 3-level pyramidal LK, 5x5 window, 3
iterations per level.  One "step" = one pass of the hot path over one batch of
``--pairs`` synthetic frame pairs per GPU, inputs already resident in HBM.
Frame pairs are independent units, so ranks share nothing on the data path
(weak scaling: pairs per GPU fixed); the only collective is the timing MAX.

Extra objects in the JSON line:
  roofline     -- dominant kernel (fused LK iteration at the finest level):
                  algorithmic bytes per launch / average launch duration measured
                  with HIP events on the launch stream during the timed region
  cpu_baseline -- the CPU oracle (a bit-exact port of the reference's NumPy/SciPy
                  arithmetic) timed on this host on a bounded sample (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_pair(dims, levels: int, iters_run) -> dict:
    """SURVEY.md section 8d byte model, per frame pair, per kernel class.

    dims: [(H_l, W_l)] coarse -> fine; iters_run[l]: iterations executed at level l.
    """
    n = [h * w for h, w in dims]
    pyr = sum(2 * (4 * n[l + 1] + 4 * n[l]) for l in range(levels - 1))      # both frames: read fine, write coarse
    it = sum(int(iters_run[l]) * 24 * n[l] for l in range(levels))            # prev, curr, u, v in; u, v out
    ups = sum(8 * n[l - 1] + 8 * n[l] for l in range(1, levels))             # u, v coarse in; u, v fine out
    return {"pyramid": pyr, "iterations": it, "upsample": ups, "total": pyr + it + ups,
            "finest_iteration_launch": 24 * n[-1]}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--cpu-sample-pairs", type=int, default=16, help="1080p pairs the CPU oracle is timed on (~10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-pair", action="store_true", help="skip the informational batch-of-1 timing")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: liboflk then binds to the HIP runtime torch already loaded

    from oflk_dist import Group, env_rank

    rank, local_rank, world = env_rank()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = Group("nccl", dev)  # backend "nccl" is RCCL on ROCm; no-op for one rank

    import _oflk
    from oflk_synth import synth_pair

    B, H, W, L, K = args.pairs, args.height, args.width, args.levels, args.iters
    # a few distinct synthetic pairs per rank, tiled to the batch (different ranks get different pairs)
    n_distinct = min(B, 4)
    host = [synth_pair(H, W, pair_index=rank * n_distinct + i) for i in range(n_distinct)]
    prev = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    curr = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    for b in range(B):
        p, c = host[b % n_distinct]
        prev[b].copy_(torch.from_numpy(p))
        curr[b].copy_(torch.from_numpy(c))
    u = torch.empty_like(prev)
    v = torch.empty_like(prev)
    plan = _oflk.Plan(local_rank, B, H, W, L, args.window, K)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        group.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # timed region: HIP event pairs (on the launch stream) around the dominant kernel only --
    # bracketing all 15 launches of a step costs ~3 % of the step time
    plan.set_profiling(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    fence()
    elapsed = group.max_over_ranks(elapsed)
    dom = plan.kernel_times().get("lk_iter_finest", {"total_ms": 0.0, "launches": 0})
    # informational per-kernel breakdown from a few extra, untimed steps with every launch bracketed
    plan.set_profiling(1)
    for _ in range(min(args.steps, 5)):
        step()
    torch.cuda.synchronize()
    ktimes = plan.kernel_times()
    plan.set_profiling(0)
    log, runs = plan.read_log(stream)

    # job-wide result summary: one small SUM all-reduce (RCCL) of per-rank totals, outside the timed region
    tot_u = group.sum_over_ranks(float(u.abs().sum(dtype=torch.float64).item()))
    tot_v = group.sum_over_ranks(float(v.abs().sum(dtype=torch.float64).item()))
    tot_pairs = int(round(group.sum_over_ranks(float(B))))
    job_stats = {"pairs_per_step": tot_pairs, "mean_abs_u": round(tot_u / (tot_pairs * H * W), 6),
                 "mean_abs_v": round(tot_v / (tot_pairs * H * W), 6)}

    total_pix = float(world) * B * H * W * args.steps
    value = total_pix / elapsed / 1e6

    # ---- roofline of the dominant kernel ------------------------------------
    import lucas_kanade_pyramidal as P

    dims = P.pyramid_level_shapes((H, W), L)
    model = algorithmic_bytes_per_pair(dims, L, runs[0])
    roofline = None
    if dom["launches"]:
        avg_ms = dom["total_ms"] / dom["launches"]
        bytes_per_launch = model["finest_iteration_launch"] * B
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from PMC counters are collected by tools/measure_traffic.sh in
        # separate rocprofv3 passes (they cannot be read live); used when they match this workload
        traffic = None
        for f in sorted((ROOT / "profiles").glob("*_hbm_traffic.json")):
            try:
                tr = json.loads(f.read_text())
                if tr.get("pairs") == B and tr.get("shape") == [H, W]:
                    traffic = tr["hbm_bytes_per_launch"]
            except Exception:
                pass
        roofline = {"bound": "hbm", "kernel": "k_lkw<2,ITER> (fused LK iteration, finest level)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": bytes_per_launch}
    # whole-call view: algorithmic bytes of the full pyramidal call over step time
    step_bytes = sum(algorithmic_bytes_per_pair(dims, L, runs[b])["total"] for b in range(B))
    whole = {"algorithmic_bytes_per_step": step_bytes,
             "achieved_GBs": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
             "frac_of_peak": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    kernels = {k: {"avg_us": round(1e3 * t["total_ms"] / t["launches"], 2), "launches": t["launches"],
                   "share": round(t["total_ms"] / max(sum(x["total_ms"] for x in ktimes.values()), 1e-12), 4)}
               for k, t in ktimes.items() if t["launches"]}

    # ---- one pair per call (BASELINE config 3 read literally): latency-bound, informational ----
    one_pair = None
    if rank == 0 and world == 1 and not args.no_one_pair:
        plan1 = _oflk.Plan(local_rank, 1, H, W, L, args.window, K)
        for _ in range(5):
            plan1.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        nrep = 100
        for _ in range(nrep):
            plan1.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        torch.cuda.synchronize()
        q1 = time.perf_counter()
        plan1.close()
        one_pair = {"us_per_call": round(1e6 * (q1 - q0) / nrep, 1), "Mpix/s": round(nrep * H * W / (q1 - q0) / 1e6, 1),
                    "note": "batch of 1: 15 dependent kernel launches, not the throughput figure"}

    # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1) --------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "oracle"))
        import oflk_oracle as O  # checker/baseline only; never on the measured path

        O.set_threads(1)
        n = max(1, args.cpu_sample_pairs)
        c0 = time.perf_counter()
        for i in range(n):
            p, c = host[i % n_distinct]
            O.lucas_kanade_pyramidal(p, c, L, args.window, K)
        c1 = time.perf_counter()
        cpu = {"value": round(n * H * W / (c1 - c0) / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
               "sample": f"{n} pairs {W}x{H}, {L}-level pyramidal {args.window}x{args.window} x{K} iters, "
                         f"oracle/oflk_oracle.c single thread, {c1 - c0:.1f} s"}
        nt = min(O.max_threads(), len(os.sched_getaffinity(0)), 16)  # the box's CPU share for one GPU
        if nt > 1:
            O.set_threads(nt)
            c0 = time.perf_counter()
            for i in range(n):
                p, c = host[i % n_distinct]
                O.lucas_kanade_pyramidal(p, c, L, args.window, K)
            c1 = time.perf_counter()
            cpu["all_cores"] = {"value": round(n * H * W / (c1 - c0) / 1e6, 4), "cores": nt}
            O.set_threads(1)

    if rank == 0:
        out = {
            "metric": "Mpix/s dense flow (1080p pyramidal)" if (H, W) == (1080, 1920) else "Mpix/s dense flow",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} frame pairs, {L}-level pyramidal LK, {args.window}x{args.window} window, "
                                   f"{K} iterations/level", "pairs_per_gpu_per_step": B,
                       "parallelism": f"frame-pair sharding x{world}", "iterations_run_pair0": [int(x) for x in runs[0]]},
            "roofline": roofline,
            "whole_call": whole,
            "kernels": kernels,
            "job_stats": job_stats,
            "one_pair_per_call": one_pair,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    plan.close()
    group.close()


if __name__ == "__main__":
    main()

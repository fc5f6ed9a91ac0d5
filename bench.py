#!/usr/bin/env python3
"""bench.py -- Mpix/s of dense pyramidal Lucas-Kanade flow on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  Rank 0 prints
ONE JSON line.  Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment),
bench.py launches those N ranks itself: the parent starts ``python -m torch.distributed.run
--nproc-per-node N bench.py ...`` as a CHILD process before anything touches the GPU, relays
rank 0's JSON line and the child's exit code.  ``--multi inproc`` is the other multi-GPU
form of the same job: ONE process, one plan per GPU on its shard of the pairs (what
oflk_pyramidal_batch_multi does inside the library, here with device-resident shards).

Workloads (``--config``):
  1080p (default)  BASELINE.json configs[2], the config the metric is quoted on: batches of
                   independent 1920x1080 frame pairs, 3-level pyramidal LK, 5x5 window, 3 iterations per
                   level; ``--pairs`` pairs per GPU per step, so N GPUs do N times the work ("weak").
  4k64             BASELINE.json configs[3]: ONE job of 64 pairs of 3840x2160, cut over the ranks with
                   shard_range(64, rank, N) -- the same job at N = 1/2/4/8 ("strong").
One "step" = one pass of the hot path over the rank's pairs, inputs already resident in HBM.
Frame pairs are independent units: ranks share nothing on the data path; the only collectives are
the barrier of the fence, the MAX of the elapsed time and one SUM of per-rank result totals
(RCCL, backend "nccl"), all outside or around the timed region (oflk_dist.run_timed).

Extra objects in the JSON line:
  roofline      dominant kernel (fused LK iteration at the finest level): algorithmic bytes per
                launch / average launch duration from HIP events on the launch stream during the
                timed region; plus the instruction-side bounds of that kernel (valu_pipe,
                issue_cadence: tools/issue_bounds.py) when profiles/ holds them for this workload
  roofline_pyr  the same for the fused pyramid kernel (fp64-bound by SciPy's arithmetic)
  cpu_baseline  the CPU oracle (a bit-exact port of the reference's NumPy/SciPy arithmetic) timed on
                this host on a bounded sample (N = 1 only)
  epe_vs_reference  the parity half of the metric: the 13 committed verification patterns through the HIP
                path (outside the timed region), compared with the digests / dense flows the reference produced
                (tests/golden/: fixtures are data; no oracle involved)
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
            "finest_iteration_launch": 24 * n[-1],
            "finest_pyramid_launch": (2 * (4 * n[-1] + 4 * n[-2])) if levels > 1 else 0}


def profile_file(pattern: str, pairs: int, shape, any_pairs: bool = False) -> dict | None:
    """The newest profiles/<tag>_<pattern>.json measured on this workload (tools/measure_traffic.sh,
    tools/issue_bounds.py).  any_pairs: accept a file measured with another batch size of the same frame shape (the
    caller scales per-launch quantities by the pair ratio: instruction counts of a launch are per pair)."""
    best = None
    for f in sorted((ROOT / "profiles").glob(f"*_{pattern}.json")):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        if (any_pairs or d.get("pairs") == pairs) and d.get("shape") == list(shape):
            best = dict(d, _file=f.name)
    return best


def live_traffic(pairs: int, shape, timeout_s: int = 240) -> dict | None:
    """HBM bytes per launch of the dominant kernel, measured during this bench run: two child processes run the same
    workload (tools/kbench.py: same plan, same synthetic frames) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
    `... --pmc WRITE_SIZE` (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; counters in units of
    1024 B, FETCH_SIZE doubled on gfx950).  Returns None when rocprofv3 is missing, when this process is itself being
    profiled, or when a pass fails -- the caller then falls back to the committed profiles/ measurement."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    out = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            # separate passes (the guide: FETCH_SIZE and WRITE_SIZE do not fit one); the third one counts the launch's
            # vector-ALU wave-instructions for roofline.valu_pipe
            for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")):
                d = os.path.join(tmp, counters[0])
                cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
                       sys.executable, str(ROOT / "tools" / "kbench.py"), "--pairs", str(pairs), "--height", str(shape[0]),
                       "--width", str(shape[1]), "--reps", "3"]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, timeout=timeout_s)
                if r.returncode != 0:
                    if counters[0] == "SQ_INSTS_VALU":
                        break   # the traffic figure stands without it
                    return None
                for counter in counters:
                    rows = []
                    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                        for row in csv.DictReader(open(f)):
                            if "k_lkw<2, 1" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                                rows.append((int(row["Grid_Size"]), float(row["Counter_Value"]),
                                             (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
                    if not rows:
                        if counters[0] == "SQ_INSTS_VALU":
                            continue
                        return None
                    big = max(g for g, _, _ in rows)   # finest level = largest grid
                    vals = [v for g, v, _ in rows if g == big]
                    out[counter] = (sum(vals) / len(vals), len(vals), sum(u for g, _, u in rows if g == big) / len(vals))
    except Exception:
        return None
    if "FETCH_SIZE" not in out or "WRITE_SIZE" not in out:
        return None
    fetch, write = out["FETCH_SIZE"][0] * 1024 * 2, out["WRITE_SIZE"][0] * 1024
    res = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
           "launches_averaged": [out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]]}
    if "SQ_INSTS_VALU" in out and "GRBM_GUI_ACTIVE" in out:
        us = out["SQ_INSTS_VALU"][2]
        res["valu_wave_instructions_per_launch"] = out["SQ_INSTS_VALU"][0]
        res["launch_us_under_pmc"] = us
        res["clock_GHz_under_pmc"] = out["GRBM_GUI_ACTIVE"][0] / 8.0 / (us * 1e3)   # the counter is summed over the 8 XCDs
    return res


def self_launch(argv) -> int:
    """`python bench.py --gpus N` (N > 1) started without a launcher: run the N ranks as a child
    `python -m torch.distributed.run` job, relay its output.  The parent never touches the GPU."""
    import socket
    import subprocess

    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    with socket.socket() as sk:   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    json_lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if json_lines:
        print(json_lines[-1])
    return r.returncode


def epe_vs_reference() -> dict | None:
    """BASELINE metric, second half ("+ EPE vs Python ref"): the 13 verification patterns (tests/golden/
    patterns_320x240.npz) through the HIP path, single-scale and pyramidal, against what the reference produced on
    them: sha256 digests of its 26 flow fields (reference_13patterns.json) and its dense translate_medium flows."""
    import hashlib

    import numpy as np

    gold = ROOT / "tests" / "golden"
    try:
        z = np.load(gold / "patterns_320x240.npz")
        ref = json.loads((gold / "reference_13patterns.json").read_text())["patterns"]
        dense = np.load(gold / "dense_translate_medium.npz")
    except Exception:
        return None
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P

    def digest(a):
        return hashlib.sha256((np.ascontiguousarray(a, np.float32) + np.float32(0.0)).tobytes()).hexdigest()

    equal, total, iters_equal, worst = 0, 0, 0, 0.0
    p = z["frame_0"].astype(np.float32)
    for name, r in ref.items():
        c = z[f"frame_1__{name}"].astype(np.float32)
        u, v = K.lucas_kanade_single_scale(p, c, 5)
        ok_s = digest(u) == r["single_scale"]["u_sha256"] and digest(v) == r["single_scale"]["v_sha256"]
        pu, pv, _, runs = P.lucas_kanade_pyramidal_with_log(p, c, 3, 5, 3)
        ok_p = digest(pu) == r["pyramidal"]["u_sha256"] and digest(pv) == r["pyramidal"]["v_sha256"]
        equal += int(ok_s) + int(ok_p)
        total += 2
        iters_equal += int(list(runs) == r["pyramidal"]["iters_run"])
        if name == "translate_medium":
            for (a, b2, ku, kv) in ((u, v, "single_u", "single_v"), (pu, pv, "pyr_u", "pyr_v")):
                worst = max(worst, float(np.mean(np.sqrt((a.astype(np.float64) - dense[ku]) ** 2 + (b2.astype(np.float64) - dense[kv]) ** 2))))
    # equal digests = the reference's flow value for value: EPE exactly 0 on that field
    return {"patterns": len(ref), "flow_fields": total, "digests_equal": equal, "iteration_counts_equal": iters_equal,
            "max_mean_epe": worst if equal == total else None, "mean_epe_dense_translate_medium": worst, "tolerance": 1e-4,
            "source": "tests/golden/reference_13patterns.json, dense_translate_medium.npz (made by importing the reference)"}


def run_inproc(args) -> None:
    """--multi inproc: the same job in ONE process, a plan per GPU on its shard of the pairs (device-resident), every
    step enqueued on all GPUs before any is waited for.  value = job pixels / wall time over all GPUs."""
    import numpy as np
    import torch

    import _oflk
    from oflk_dist import job_layout
    from oflk_synth import synth_pair

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path exists)")
    n = args.gpus
    have = torch.cuda.device_count()
    if args.force_device is None and n > have:
        raise SystemExit(f"--gpus {n} but {have} GPU(s) visible")
    L, K = args.levels, args.iters
    shards = []
    host = None
    for r in range(n):
        lay = job_layout(args.config, r, n, args.pairs, args.height, args.width)
        d = args.force_device if args.force_device is not None else r
        dev = torch.device("cuda", d)
        if host is None:
            host = [synth_pair(lay.height, lay.width, pair_index=i) for i in range(2)]
        with torch.cuda.device(dev):
            prev = torch.empty((lay.pairs_local, lay.height, lay.width), dtype=torch.float32, device=dev)
            curr = torch.empty_like(prev)
            hp = [(torch.from_numpy(p).to(dev), torch.from_numpy(c).to(dev)) for p, c in host]
            for b in range(lay.pairs_local):
                p, c = hp[(lay.pair_begin + b) % len(hp)]
                prev[b].copy_(p)
                curr[b].copy_(c)
            del hp
            u, v = torch.empty_like(prev), torch.empty_like(prev)
            stream = torch.cuda.Stream(device=dev)
            plan = _oflk.Plan(d, lay.pairs_local, lay.height, lay.width, L, args.window, K)
        shards.append((lay, dev, prev, curr, u, v, stream, plan))

    def step():
        for lay, dev, prev, curr, u, v, stream, plan in shards:
            plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream.cuda_stream)

    def sync():
        for sh in shards:
            sh[6].synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    lay0 = shards[0][0]
    H, W = lay0.height, lay0.width
    pairs = sum(sh[0].pairs_local for sh in shards)
    abs_u = sum(float(sh[4].abs().sum(dtype=torch.float64).item()) for sh in shards)
    out = {"metric": "Mpix/s dense flow (1080p pyramidal)" if (H, W) == (1080, 1920) else "Mpix/s dense flow",
           "value": round(pairs * H * W * args.steps / elapsed / 1e6, 1), "unit": "Mpix/s", "n_gpus": n, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
           "scaling": lay0.scaling, "vs_baseline": None, "dtype": "f32",
           "data": "synthetic" if args.force_device is None else "synthetic (REHEARSAL: not a measurement)",
           "config": {"workload": f"{W}x{H} frame pairs, {L}-level pyramidal LK, {args.window}x{args.window} window, {K} iterations/level",
                      "name": lay0.config, "what": lay0.label, "pairs_per_step_job": pairs,
                      "pairs_per_gpu_per_step": [sh[0].pairs_local for sh in shards],
                      "parallelism": f"frame-pair sharding x{n}, one process, a plan and a stream per GPU (--multi inproc)"},
           "job_stats": {"pairs_per_step": pairs, "mean_abs_u": round(abs_u / (pairs * H * W), 6)},
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out))
    for sh in shards:
        sh[7].close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="1080p", choices=["1080p", "4k64"])
    ap.add_argument("--pairs", type=int, default=None, help="frame pairs per GPU per step (default: the config's)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--cpu-sample-pairs", type=int, default=16, help="1080p pairs the CPU oracle is timed on (~10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-pair", action="store_true", help="skip the informational batch-of-1 timing")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes; take roofline.traffic from profiles/")
    # rehearsal switches (a one-GPU box cannot host two RCCL ranks): gloo for the three small collectives, and
    # every rank on one device -- exercises the N > 1 code path end to end; the numbers of such a run mean nothing
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--force-device", type=int, default=None, help="use this GPU on every rank (rehearsal only)")
    ap.add_argument("--multi", default="ranks", choices=["ranks", "inproc"],
                    help="N > 1: one process per GPU (default) or ONE process with a plan per GPU")
    ap.add_argument("--no-parity", action="store_true", help="skip the 13-pattern EPE check after the timed region")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region: gather the flow shards of all ranks to rank 0 through the process group "
                         "(RCCL on device tensors) and report gather_ms / gather_GBs next to, never inside, `value`")
    args = ap.parse_args()
    if args.gpus > 1 and args.multi == "ranks" and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:]))   # before torch / the GPU are touched
    if args.gpus > 1 and args.multi == "inproc":
        run_inproc(args)
        return

    import numpy as np
    import torch  # first: liboflk then binds to the HIP runtime torch already loaded

    from oflk_dist import Group, env_rank, job_layout, job_throughput, run_timed

    rank, local_rank, world = env_rank()
    if world != args.gpus:
        # the JSON line must describe the job that was asked for: never continue with another size
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N ranks with torch.distributed.run "
              f"(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path exists)")
    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # backend "nccl" is RCCL on ROCm (device tensors); gloo reduces host tensors; no-op for one rank
    group = Group(args.backend, dev if args.backend == "nccl" else None, always=args.gather)   # --gather: RCCL even for one rank

    import _oflk
    from oflk_synth import synth_pair

    lay = job_layout(args.config, rank, world, args.pairs, args.height, args.width)
    B, H, W, L, K = lay.pairs_local, lay.height, lay.width, args.levels, args.iters
    if B < 1:
        raise SystemExit(f"rank {rank}: no pairs to process ({lay.pairs_total} pairs over {world} ranks)")
    # a few distinct synthetic pairs, tiled over the job by global pair index (so the job's data does not
    # depend on how it is cut over the ranks)
    n_distinct = 4 if H * W <= 1920 * 1080 else 2
    host = [synth_pair(H, W, pair_index=i) for i in range(n_distinct)]
    prev = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    curr = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    dev_pairs = [(torch.from_numpy(p).to(dev), torch.from_numpy(c).to(dev)) for p, c in host]
    for b in range(B):
        p, c = dev_pairs[(lay.pair_begin + b) % n_distinct]
        prev[b].copy_(p)
        curr[b].copy_(c)
    del dev_pairs
    u = torch.empty_like(prev)
    v = torch.empty_like(prev)
    plan = _oflk.Plan(local_rank, B, H, W, L, args.window, K)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)

    # timed region: HIP event pairs (on the launch stream) around the dominant kernel only --
    # bracketing all 15 launches of a step costs ~3 % of the step time
    elapsed = run_timed(group, step, torch.cuda.synchronize, args.steps, args.warmup,
                        before_timed=lambda: plan.set_profiling(2))
    dom = plan.kernel_times().get("lk_iter_finest", {"total_ms": 0.0, "launches": 0})
    # informational per-kernel breakdown from a few extra, untimed steps with every launch bracketed
    plan.set_profiling(1)
    for _ in range(min(args.steps, 5)):
        step()
    torch.cuda.synchronize()
    ktimes = plan.kernel_times()
    plan.set_profiling(0)
    log, runs = plan.read_log(stream)
    uncertain = int(plan.read_uncertain(stream).astype(bool).sum())

    # job-wide figures: pixels of all ranks / MAX-rank time; one SUM all-reduce (RCCL) of per-rank result totals
    job = job_throughput(group, lay, args.steps, elapsed,
                         {"abs_u": float(u.abs().sum(dtype=torch.float64).item()),
                          "abs_v": float(v.abs().sum(dtype=torch.float64).item())})
    value = job["Mpix_per_s"]
    npix_job = job["pairs_per_step"] * H * W
    job_stats = {"pairs_per_step": job["pairs_per_step"], "mean_abs_u": round(job["sums"]["abs_u"] / npix_job, 6),
                 "mean_abs_v": round(job["sums"]["abs_v"] / npix_job, 6)}

    # ---- roofline of the dominant kernel, and of the pyramid kernel ------------------
    import lucas_kanade_pyramidal as P

    dims = P.pyramid_level_shapes((H, W), L)
    model = algorithmic_bytes_per_pair(dims, L, runs[0])
    roofline = None
    if dom["launches"]:
        avg_ms = dom["total_ms"] / dom["launches"]
        bytes_per_launch = model["finest_iteration_launch"] * B
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        tr = profile_file("hbm_traffic", B, (H, W))
        live = None
        if world == 1 and not args.no_live_traffic and args.levels == 3 and args.window == 5 and args.iters == 3:
            live = live_traffic(B, (H, W))
        ib = profile_file("issue_bounds", B, (H, W), any_pairs=True)
        roofline = {"bound": "hbm", "binding_pipe": "valu", "timed_with_profiling": 2,
                    "kernel": "k_lkw<2,ITER> (fused LK iteration, finest level)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": live["hbm_bytes_per_launch"] if live else (tr["hbm_bytes_per_launch"] if tr else None),
                    "traffic_source": ("measured during this run: child processes of bench.py ran the same workload (tools/kbench.py) "
                                       "under rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, "
                                       f"FETCH_SIZE x2 on gfx950); launches averaged {live['launches_averaged']}") if live else
                                      ((f"profiles/{tr['_file']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
                                        f"over this command (tools/measure_traffic.sh); the live passes of this run were skipped or failed") if tr else None),
                    "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": bytes_per_launch,
                    "hbm_floor_us": round(bytes_per_launch / HBM_PEAK_GBS / 1e3, 1)}
        if ib:
            # the instruction-side bounds of the same launch (the kernel's arithmetic is the reference's, op for op):
            # dynamic instruction counts (SQ counters of a builder-side run) priced with the wall-clock-validated cycle
            # table (tools/ubench/valu_wall.hip), scaled to this batch size and set against THIS run's launch time
            scale = B / float(ib["pairs"])   # the counters were taken on a launch of ib["pairs"] pairs; instructions are per pair
            for k in ("valu_pipe", "issue_cadence"):
                roofline[k] = {"bound": k, "floor_us": round(ib[k]["floor_us"] * scale, 1),
                               "frac": round(ib[k]["floor_us"] * scale / (avg_ms * 1e3), 4)}
            if live and "valu_wave_instructions_per_launch" in live:
                # the instruction count of THIS run's launches (third rocprofv3 --pmc child pass), priced with the
                # profile's mean cycles per instruction of the kernel's static mix, at the clock of that pass
                n_valu = live["valu_wave_instructions_per_launch"]
                floor_us = n_valu * ib["mean_saturated_cycles_per_valu_instruction"] / 1024.0 / live["clock_GHz_under_pmc"] / 1e3
                roofline["valu_pipe"] = {"bound": "valu_pipe", "floor_us": round(floor_us, 1), "frac": round(floor_us / live["launch_us_under_pmc"], 4),
                                         "valu_wave_instructions_per_launch": n_valu,
                                         "source": "SQ_INSTS_VALU of this run's launches (rocprofv3 --pmc child pass) x the kernel's mean "
                                                   f"{ib['mean_saturated_cycles_per_valu_instruction']} cycles per instruction (static mix priced with the "
                                                   "wall-clock-validated table) / 1024 SIMDs / the pass's clock, against the launch time of that pass"}
                scale = None
            n_show = live["valu_wave_instructions_per_launch"] if scale is None else ib["wave_instructions_per_launch"]["valu"] * scale
            roofline["binding"] = (f"vector ALU: the launch's {n_show:.3g} VALU wave-instructions need "
                                   f"{roofline['valu_pipe']['frac']:.2f} of its time at their saturated rates ({ib.get('price_classes_cycles')} SIMD cycles per "
                                   "instruction by class, checked against wall-clock); the HBM stream needs "
                                   f"{roofline['frac']:.2f} (0.73 of the peak is what a 2:1 read:write stream reaches on this chip). The arithmetic is the "
                                   "reference's op for op (fp64 warp, NumPy-order sums, IEEE divisions), so the 0.70-of-HBM target is out of reach at EPE = 0; DESIGN.md section 5")
            roofline["issue_bounds_source"] = (f"profiles/{ib['_file']} (tools/issue_bounds.py; SQ counters of a {ib['pairs']}-pair "
                                               f"launch scaled to {B} pairs)")
    roofline_pyr = None
    pk = ktimes.get("pyr_down_fused")
    if pk and pk["launches"] and L > 1:
        # one launch per pyramid step; the finest one (both frames of every pair) carries 80 % of the bytes
        per_call_ms = (L - 1) * pk["total_ms"] / pk["launches"]   # L-1 launches per call
        pyr_bytes = algorithmic_bytes_per_pair(dims, L, runs[0])["pyramid"] * B
        a = pyr_bytes / (per_call_ms * 1e-3) / 1e9
        roofline_pyr = {"bound": "hbm", "kernel": "k_pyr_down (fused 17-tap blur x2 + linspace resample), all levels of a call",
                        "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                        "us_per_call": round(per_call_ms * 1e3, 1), "algorithmic_bytes_per_call": pyr_bytes,
                        "note": "bound by SciPy's fp64 arithmetic (25 fp64 operations per blurred pixel and axis), not by HBM"}
        ibp = (profile_file("issue_bounds", B, (H, W), any_pairs=True) or {}).get("pyr_down")
        if ibp and "valu_pipe" in ibp:
            # the finest pyramid launch alone: its vector-ALU floor against its own duration in the builder-side counter run
            roofline_pyr["valu_pipe"] = {"bound": "valu_pipe", "frac_of_launch": ibp["valu_pipe"]["frac"],
                                         "mean_cycles_per_valu_instruction": ibp["mean_saturated_cycles_per_valu_instruction"],
                                         "note": "fp64 instructions issue at 4.15 SIMD cycles each (wall-clock-validated); this kernel is at its vector-ALU floor"}
    # whole-call view: algorithmic bytes of the full pyramidal call over step time
    step_bytes = sum(algorithmic_bytes_per_pair(dims, L, runs[b])["total"] for b in range(B))
    whole = {"algorithmic_bytes_per_step": step_bytes,
             "achieved_GBs": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
             "frac_of_peak": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    kernels = {k: {"avg_us": round(1e3 * t["total_ms"] / t["launches"], 2), "launches": t["launches"],
                   "share": round(t["total_ms"] / max(sum(x["total_ms"] for x in ktimes.values()), 1e-12), 4)}
               for k, t in ktimes.items() if t["launches"]}

    # ---- the opt-in contracted arithmetic (fused multiply-adds in the Gaussian pyramid; include/oflk.h): informational,
    # never `value` -- what it buys on this workload and how many flow values it changes against the exact result ----
    contracted = None
    if rank == 0 and world == 1 and L > 1:
        u_ex, v_ex = u.clone(), v.clone()
        plan.set_arithmetic(1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        nrep = max(3, min(args.steps, 10))
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        c1 = time.perf_counter()
        plan.set_arithmetic(0)
        differing = int(((u != u_ex) | (v != v_ex)).sum().item())
        d2 = (u.double() - u_ex.double()) ** 2 + (v.double() - v_ex.double()) ** 2
        contracted = {"value": round(nrep * B * H * W / (c1 - c0) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(1e3 * (c1 - c0) / nrep, 4),
                      "flow_values_differing_from_exact": differing, "of": int(2 * u.numel()),
                      "mean_epe_vs_exact": float(d2.sqrt().mean().item()),
                      "note": "oflk_plan_set_arithmetic(OFLK_ARITH_CONTRACTED): opt-in, not the metric's value"}
        del u_ex, v_ex, d2
        step()   # leave the exact result in u, v
        torch.cuda.synchronize()

    # ---- the opt-in within-tolerance arithmetic (oflk_plan_set_arithmetic(OFLK_ARITH_TOLERANT); include/oflk.h): never `value`.
    # Timed like the exact mode (HIP events around the dominant kernel only), then graded against the REFERENCE's own dense
    # flows (tests/golden/dense_reference_flows.npz, made by importing it): the 13 verification patterns and pair 0 of this
    # workload; the bar is the north star's mean endpoint error <= 1e-4 px per field ----
    tolerant = None
    if rank == 0 and world == 1 and L > 1 and args.window in (4, 5):
        plan.set_arithmetic(2)
        for _ in range(2):
            step()
        plan.set_profiling(2)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        nrep = max(3, min(args.steps, 10))
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        c1 = time.perf_counter()
        kt = plan.kernel_times().get("lk_iter_finest", {"total_ms": 0.0, "launches": 0})
        plan.set_profiling(0)
        tol_ms = 1e3 * (c1 - c0) / nrep
        tolerant = {"value": round(nrep * B * H * W / (c1 - c0) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(tol_ms, 4),
                    "timed_with_profiling": 2,
                    "note": "oflk_plan_set_arithmetic(OFLK_ARITH_TOLERANT): opt-in, not the metric's value. Pyramid with fused multiply-adds; the "
                            "iterations of the two finest levels in the streaming kernel k_lks (fused-lerp fp64 warp, window sums vertical-then-"
                            "horizontal instead of NumPy's order); coarser levels, solve and flow upsampling exact"}
        if kt["launches"]:
            t_ms = kt["total_ms"] / kt["launches"]
            bpl = model["finest_iteration_launch"] * B
            tolerant["roofline"] = {"bound": "hbm", "binding_pipe": "hbm", "kernel": "k_lks<ITER> (streaming LK iteration, finest level)",
                                    "achieved": round(bpl / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(bpl / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_us": round(t_ms * 1e3, 2),
                                    "algorithmic_bytes_per_launch": bpl, "traffic": None}
            tolerant["roofline"]["note"] = ("average over the level's three launches; the first one also does the flow upsampling "
                                            "(SURVEY section 8d counts it separately, 10 B/px) while reading 16 instead of 24 B/px: priced at 24 like the others")
            tolerant["whole_call"] = {"algorithmic_bytes_per_step": step_bytes, "achieved_GBs": round(step_bytes / (tol_ms * 1e-3) / 1e9, 1),
                                      "frac_of_peak": round(step_bytes / (tol_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            trt = profile_file("hbm_traffic_tolerant", B, (H, W))
            if trt:
                tolerant["roofline"]["traffic"] = trt["hbm_bytes_per_launch"]
                tolerant["roofline"]["traffic_source"] = (f"profiles/{trt['_file']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                                                         "this command (tools/profiles_r04.sh)")
        try:
            dense = np.load(ROOT / "tests" / "golden" / "dense_reference_flows.npz")
            z = np.load(ROOT / "tests" / "golden" / "patterns_320x240.npz")
            names = [k[len("frame_1__"):] for k in z.files if k.startswith("frame_1__")]

            def mean_epe(uu, vv, ru, rv):
                return float(np.mean(np.sqrt((uu.astype(np.float64) - ru) ** 2 + (vv.astype(np.float64) - rv) ** 2)))

            epes = {}
            if (H, W, L, K) == (1080, 1920, 3, 3) and lay.pair_begin == 0:
                epes["workload_pair0"] = mean_epe(u[0].cpu().numpy(), v[0].cpu().numpy(), dense["bench_1080p_pair0__u"], dense["bench_1080p_pair0__v"])
            pp = torch.from_numpy(np.stack([z["frame_0"].astype(np.float32)] * len(names))).to(dev)
            cc = torch.from_numpy(np.stack([z[f"frame_1__{n}"].astype(np.float32) for n in names])).to(dev)
            pu, pv = torch.empty_like(pp), torch.empty_like(pp)
            plan13 = _oflk.Plan(local_rank, len(names), 240, 320, 3, 5, 3)
            plan13.set_arithmetic(2)
            plan13.pyramidal(pp.data_ptr(), cc.data_ptr(), pu.data_ptr(), pv.data_ptr(), stream)
            torch.cuda.synchronize()
            plan13.close()
            hu, hv = pu.cpu().numpy(), pv.cpu().numpy()
            for i, n in enumerate(names):
                epes[n] = mean_epe(hu[i], hv[i], dense[f"{n}__u"], dense[f"{n}__v"])
            worst = max(epes, key=epes.get)
            tolerant["max_mean_epe_vs_reference"] = epes[worst]
            tolerant["worst_field"] = worst
            tolerant["fields"] = len(epes)
            tolerant["fields_within_1e-4"] = int(sum(e <= 1e-4 for e in epes.values()))
            if "workload_pair0" in epes:
                tolerant["mean_epe_workload_pair0"] = epes["workload_pair0"]
            tolerant["reference"] = "tests/golden/dense_reference_flows.npz: dense flows of the reference itself (13 verification patterns + pair 0 of this workload)"
        except Exception as e:   # the fixture is part of the repository; a run without it still reports the timing
            tolerant["max_mean_epe_vs_reference"] = None
            tolerant["epe_error"] = repr(e)
        plan.set_arithmetic(0)
        step()   # leave the exact result in u, v
        torch.cuda.synchronize()

    # ---- BASELINE configs[1], batched: 256 pairs of 640x480, single-scale 5x5 (the streaming kernel: exact on these 8-bit
    # frames, doubtful tiles redone in NumPy's order inside the call); informational, 16 B/px algorithmic ----
    single = None
    if rank == 0 and world == 1 and not args.no_one_pair:
        try:
            hs = [synth_pair(480, 640, pair_index=i) for i in range(2)]
            sp = torch.stack([torch.from_numpy(hs[i % 2][0]) for i in range(256)]).to(dev)
            sc = torch.stack([torch.from_numpy(hs[i % 2][1]) for i in range(256)]).to(dev)
            su_, sv_ = torch.empty_like(sp), torch.empty_like(sp)
            plan_s = _oflk.Plan(local_rank, 256, 480, 640, 1, 5, 0)
            for _ in range(3):
                plan_s.single_scale(sp.data_ptr(), sc.data_ptr(), su_.data_ptr(), sv_.data_ptr(), stream)
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for _ in range(20):
                plan_s.single_scale(sp.data_ptr(), sc.data_ptr(), su_.data_ptr(), sv_.data_ptr(), stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - s0) / 20
            plan_s.close()
            npx = 256 * 480 * 640
            single = {"workload": "256 pairs of 640x480, single-scale 5x5 (BASELINE configs[1], batched)", "us_per_call": round(dt * 1e6, 1),
                      "Mpix/s": round(npx / dt / 1e6, 1), "algorithmic_GBs": round(npx * 16 / dt / 1e9, 1),
                      "frac_of_hbm_peak": round(npx * 16 / dt / 1e9 / HBM_PEAK_GBS, 4),
                      "kernel": "k_lks<SINGLE> + redo pass of k_lkw<2, SINGLE> (exact; rounds 1 - 3 ran the tile kernel throughout: 0.337)"}
            del sp, sc, su_, sv_
        except Exception as e:   # informational leg: never fails the bench
            single = {"error": repr(e)}

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
                    "note": "batch of 1: 14 dependent kernel launches, not the throughput figure"}

    # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1) --------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "oracle"))
        import oflk_oracle as O  # checker/baseline only; never on the measured path

        O.set_threads(1)
        # ~10 s of single-thread work whatever the frame size
        n = max(1, int(round(args.cpu_sample_pairs * (1080 * 1920) / float(H * W))))
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
        # the Python reference itself cannot travel to this box; its time on pair 0 of this workload was recorded where
        # the golden digests were made (tests/golden/make_golden_fullsize.py, the build container, one core)
        try:
            for c in json.loads((ROOT / "tests" / "golden" / "reference_fullsize.json").read_text()).values():
                if c.get("mode") == "pyramidal" and c["shape"] == [H, W] and [c["levels"], c["window_size"], c["iterations"]] == [L, args.window, K]:
                    cpu["python_reference_in_build_container"] = {"value": round(H * W / c["reference_seconds"] / 1e6, 5), "unit": "Mpix/s",
                                                                  "cores": 1, "seconds_per_pair": c["reference_seconds"]}
        except Exception:
            pass

    parity = None
    if rank == 0 and not args.no_parity:
        if local_rank != 0:
            _oflk.check(_oflk.lib().oflk_set_device(local_rank))
        parity = epe_vs_reference()
        # pair 0 of THIS workload against the reference's own output on it (tests/golden/reference_fullsize.json, made by
        # importing the reference: 3 minutes for the 1080p pair, 15 for the 4K one); u[0], v[0] as the batch plan (or the one-pair
        # plan after it: same values) left them
        try:
            full = json.loads((ROOT / "tests" / "golden" / "reference_fullsize.json").read_text())
        except Exception:
            full = {}
        for key, c in full.items():
            if (parity is not None and lay.pair_begin == 0 and c.get("mode") == "pyramidal" and c["shape"] == [H, W] and
                    [c["levels"], c["window_size"], c["iterations"]] == [L, args.window, K] and c["pair_index"] == 0):
                import hashlib

                def dg(t):
                    return hashlib.sha256((np.ascontiguousarray(t.cpu().numpy(), np.float32) + np.float32(0.0)).tobytes()).hexdigest()

                eq = int(dg(u[0]) == c["u_sha256"]) + int(dg(v[0]) == c["v_sha256"])
                parity["workload_pair0"] = {"flow_fields": 2, "digests_equal": eq, "mean_epe": 0.0 if eq == 2 else None,
                                            "source": f"tests/golden/reference_fullsize.json[{key}]: the reference's own flow of pair 0 "
                                                      f"of this workload ({c['reference_seconds']} s on one core there)"}
    # ---- optional: the flow shards to rank 0 over the process group (reported separately) ----
    gathered = None
    if args.gather:
        from oflk_dist import gather_flows

        lays = [job_layout(args.config, r, world, args.pairs, args.height, args.width) for r in range(world)]
        if args.backend == "nccl":
            g = gather_flows(group, u, v, lays, torch.cuda.synchronize)
        else:   # rehearsal: gloo moves host tensors
            g = gather_flows(group, u.cpu(), v.cpu(), lays, torch.cuda.synchronize)
        gathered = {"gather_ms": round(g["gather_ms"], 3), "gather_GBs": round(g["gather_GBs"], 2), "bytes_received_by_rank0": g["bytes_received"],
                    "pairs": g["pairs"], "backend": args.backend,
                    "note": "one gather collective of the [2][pairs][H][W] float32 shards to rank 0 after the timed region; not part of `value`"}
        if rank == 0:
            gathered["checksum_abs_u"] = float(g["u"].abs().sum(dtype=torch.float64).item())   # equals the SUM all-reduce of the ranks' totals
            gathered["equals_allreduce_total"] = bool(abs(gathered["checksum_abs_u"] - job["sums"]["abs_u"]) <= 1e-9 * max(job["sums"]["abs_u"], 1.0))
        del g
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
            "scaling": lay.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.backend == "nccl" and args.force_device is None else "synthetic (REHEARSAL: not a measurement)",
            "config": {"workload": f"{W}x{H} frame pairs, {L}-level pyramidal LK, {args.window}x{args.window} window, "
                                   f"{K} iterations/level", "name": lay.config, "what": lay.label,
                       "pairs_per_gpu_per_step": B, "pairs_per_step_job": job["pairs_per_step"],
                       "parallelism": f"frame-pair sharding x{world}", "iterations_run_pair0": [int(x) for x in runs[0]],
                       "exit_decisions_within_5e-5_of_threshold": uncertain},
            "roofline": roofline,
            "roofline_pyr": roofline_pyr,
            "whole_call": whole,
            "kernels": kernels,
            "job_stats": job_stats,
            "contracted_arithmetic": contracted,
            "tolerance_mode": tolerant,
            "single_scale_batched": single,
            "gather": gathered,
            "one_pair_per_call": one_pair,
            "cpu_baseline": cpu,
            "epe_vs_reference": parity,
        }
        print(json.dumps(out))
    plan.close()
    group.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel-level timing helper (development tool, not part of the bench contract):
per-kernel average durations from the plan's HIP-event profiling."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--identical", action="store_true", help="curr = prev (exercises early exit)")
    ap.add_argument("--graph", action="store_true", help="also time one pass captured into a HIP graph")
    ap.add_argument("--kernels", type=int, default=0, help="oflk_plan_set_kernels of the single-scale plan (0 automatic: 5x5 streams; 1: the tile kernel)")
    ap.add_argument("--only", default="", help="'single' or 'pyramidal': run only that plan")
    ap.add_argument("--arith", type=int, default=0, help="oflk_plan_set_arithmetic mode of the pyramidal plan (0 exact, 1 contracted, 2 tolerant)")
    args = ap.parse_args()
    import torch

    import _oflk
    from oflk_synth import synth_pair

    dev = torch.device("cuda", 0)
    B, H, W = args.pairs, args.height, args.width
    host = [synth_pair(H, W, i) for i in range(min(B, 4))]
    prev = torch.stack([torch.from_numpy(host[b % len(host)][0]) for b in range(B)]).to(dev)
    curr = prev.clone() if args.identical else torch.stack([torch.from_numpy(host[b % len(host)][1]) for b in range(B)]).to(dev)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    stream = torch.cuda.current_stream().cuda_stream
    npx = B * H * W
    for name, plan, fn in (
        ("single", _oflk.Plan(0, B, H, W, 1, args.window, 0), "single_scale"),
        ("pyramidal", _oflk.Plan(0, B, H, W, 3, args.window, 3), "pyramidal"),
    ):
        if args.only and args.only != name:
            plan.close()
            continue
        call = getattr(plan, fn)
        if name == "pyramidal" and args.arith:
            plan.set_arithmetic(args.arith)
        if name == "single" and args.kernels:
            plan.set_kernels(args.kernels)
        for _ in range(2):
            call(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            call(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        print(f"== {name} without per-kernel events: {e0.elapsed_time(e1) / args.reps * 1e3:.1f} us/call")
        if args.graph:
            # the plan only enqueues kernels: a caller may capture one pass into a HIP graph and replay it
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                call(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), torch.cuda.current_stream().cuda_stream)
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize()
            g0 = torch.cuda.Event(enable_timing=True)
            g1 = torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(args.reps):
                g.replay()
            g1.record()
            torch.cuda.synchronize()
            print(f"== {name} as a captured HIP graph: {g0.elapsed_time(g1) / args.reps * 1e3:.1f} us/call")
        plan.set_profiling(True)
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.reps):
            call(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / args.reps
        print(f"== {name}: {ms*1e3:.1f} us/call, {npx/ms/1e3:.0f} Mpix/s")
        for k, t in plan.kernel_times().items():
            if t["launches"]:
                avg = 1e3 * t["total_ms"] / t["launches"]
                print(f"   {k:16s} {avg:10.1f} us x {t['launches'] // args.reps}/call")
        plan.close()


if __name__ == "__main__":
    main()

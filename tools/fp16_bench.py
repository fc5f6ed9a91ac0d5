#!/usr/bin/env python3
"""Timing of the fp16 single-scale mode (BASELINE config 5 geometry by default), development tool.
Usage (GPU box): python3 tools/fp16_bench.py [--height 4320 --width 7680 --window 7 --pairs 1 --reps 20]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--window", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch

    import _oflk
    from oflk_synth import synth_pair

    dev = torch.device("cuda", 0)
    B, H, W = args.pairs, args.height, args.width
    p0, c0 = synth_pair(H, W, 0)
    prev = torch.from_numpy(p0).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    curr = torch.from_numpy(c0).to(dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    st = torch.cuda.current_stream().cuda_stream
    plan = _oflk.Plan(0, B, H, W, 1, args.window, 0)
    for _ in range(3):
        plan.single_scale_fp16(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), 255.0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        plan.single_scale_fp16(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), 255.0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    n = B * H * W
    print(f"fp16 {B} x {H}x{W} window {args.window}: {dt * 1e6:.1f} us/call, {n / dt / 1e9:.1f} Gpix/s, "
          f"{n * 16 / dt / 1e12:.2f} TB/s algorithmic ({n * 16 / dt / 8e12:.3f} of 8 TB/s)")


if __name__ == "__main__":
    main()

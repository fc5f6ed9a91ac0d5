#!/usr/bin/env python3
"""Which pipe binds the LK kernels?  (development tool)

Runs the pyramidal step (32 x 1080p by default) alone and beside a partner kernel on a second stream that
occupies exactly one resource (tools/ubench/spinner.hip):
  sleep      residency control: the partner's waves take the same wave slots / registers and issue nothing
  valu32     every partner wave issues independent v_add_f32 back to back
  valu32/2   the same, thinned with s_nop
  valu64     v_add_f64
  hbm        float4 copy over 2 x 1 GiB (far beyond the Infinity Cache)
and reports the step time, the dominant kernel's time (plan profiling mode 2) and what the partner achieved
beside the step against what it achieves alone.  The partner is launched first and outlives the timed steps.
"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch

    import _oflk
    from oflk_synth import synth_pair

    spin = ctypes.CDLL(str(ROOT / "tools" / "ubench" / "libspin.so"))
    spin.spin_valu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    spin.spin_hbm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]

    dev = torch.device("cuda", 0)
    B, H, W = args.pairs, args.height, args.width
    host = [synth_pair(H, W, i) for i in range(min(B, 4))]
    prev = torch.stack([torch.from_numpy(host[b % len(host)][0]) for b in range(B)]).to(dev)
    curr = torch.stack([torch.from_numpy(host[b % len(host)][1]) for b in range(B)]).to(dev)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    plan = _oflk.Plan(0, B, H, W, 3, 5, 3)
    sa = torch.cuda.Stream()
    sb = torch.cuda.Stream()
    nwaves_max = 256 * 8 * 4
    spin_out = torch.zeros(nwaves_max * 8, dtype=torch.int32, device=dev)
    big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB
    big_b = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    big_a.normal_()

    def step_times(n):
        """n steps on stream A: (ms per step, dominant-kernel us per launch)"""
        plan.set_profiling(2)
        plan.kernel_times()   # reset
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            e0.record(sa)
            for _ in range(n):
                plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), sa.cuda_stream)
            e1.record(sa)
        e1.synchronize()
        kt = plan.kernel_times()
        dom = [(k, t) for k, t in kt.items() if t["launches"]]
        dom_us = 1e3 * dom[0][1]["total_ms"] / dom[0][1]["launches"] if dom else float("nan")
        plan.set_profiling(0)
        return e0.elapsed_time(e1) / n, dom_us

    def partner_stats(nw):
        o = spin_out[: nw * 8].cpu().numpy().astype(np.uint32).reshape(nw, 8)
        cyc = o[:, 0].astype(np.float64) * 256.0
        rt = o[:, 1].astype(np.float64) * 10e-9   # seconds (100 MHz)
        keys = ((o[:, 4] >> 4) & 3) | (((o[:, 4] >> 8) & 0xFF) << 2) | ((o[:, 5] & 0xF) << 10)
        simds = len(np.unique(keys))
        return float(np.median(cyc)), float(np.median(rt)), simds

    for _ in range(3):
        step_times(2)
    base_ms, base_dom = step_times(args.steps)
    res = {"workload": f"{B} x {W}x{H} pyramidal 3/5/3", "alone": {"step_ms": base_ms, "dominant_us": base_dom}, "partners": []}
    print(f"alone: step {base_ms:.3f} ms, dominant kernel {base_dom:.1f} us", flush=True)

    partners = [
        # name, mode, blocks (x256 threads), nops, instr per iteration
        ("sleep, 1 wave/SIMD", 0, 256, 0, 0),
        ("sleep, 2 waves/SIMD", 0, 512, 0, 0),
        ("valu32 v_add_f32, 1 wave/SIMD", 1, 256, 0, 32),
        ("valu32 v_add_f32, 2 waves/SIMD", 1, 512, 0, 32),
        ("valu32 thinned (4 s_nop 7 per 32), 1 wave/SIMD", 1, 256, 4, 32),
        ("valu64 v_add_f64, 1 wave/SIMD", 2, 256, 0, 32),
        ("valu64 v_add_f64, 2 waves/SIMD", 2, 512, 0, 32),
    ]
    est_ms = base_ms * (args.steps + 4) * 2.5
    for name, mode, blocks, nops, ninstr in partners:
        nw = blocks * 4
        # size the partner alone first: iterations for ~est_ms
        probe = 20000
        spin.spin_valu(sb.cuda_stream, mode, blocks, 256, probe, nops, spin_out.data_ptr())
        sb.synchronize()
        cyc, rt, simds = partner_stats(nw)
        per_iter_s = rt / probe
        iters = int(est_ms * 1e-3 / per_iter_s)
        spin.spin_valu(sb.cuda_stream, mode, blocks, 256, iters, nops, spin_out.data_ptr())
        sb.synchronize()
        cyc_a, rt_a, simds_a = partner_stats(nw)
        alone_rate = iters * ninstr / cyc_a if ninstr else 0.0   # wave-instructions per cycle per wave
        # now beside the step
        spin.spin_valu(sb.cuda_stream, mode, blocks, 256, iters, nops, spin_out.data_ptr())
        time.sleep(0.003)
        step_times(2)
        ms, dom = step_times(args.steps)
        still = not sb.query()
        sb.synchronize()
        cyc_c, rt_c, simds_c = partner_stats(nw)
        co_rate = iters * ninstr / cyc_c if ninstr else 0.0
        row = {"partner": name, "step_ms": ms, "step_slowdown": ms / base_ms, "dominant_us": dom, "dominant_slowdown": dom / base_dom,
               "partner_outlived_steps": bool(still), "partner_simds": simds_c,
               "partner_instr_per_cycle_per_wave_alone": alone_rate, "partner_instr_per_cycle_per_wave_beside": co_rate,
               "partner_seconds_alone": rt_a, "partner_seconds_beside": rt_c}
        res["partners"].append(row)
        print(json.dumps(row), flush=True)

    # HBM partner
    n_vec = big_a.numel() // 4
    for blocks in (256, 1024):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        spin.spin_hbm(sb.cuda_stream, big_a.data_ptr(), big_b.data_ptr(), n_vec, 1, blocks)
        sb.synchronize()
        e0.record(sb)
        spin.spin_hbm(sb.cuda_stream, big_a.data_ptr(), big_b.data_ptr(), n_vec, 4, blocks)
        e1.record(sb)
        e1.synchronize()
        alone_s = e0.elapsed_time(e1) * 1e-3
        alone_tbs = 4 * 2 * big_a.numel() * 4 / alone_s * 1e-12
        reps = max(4, int(est_ms * 1e-3 / (alone_s / 4)))
        e0.record(sb)
        spin.spin_hbm(sb.cuda_stream, big_a.data_ptr(), big_b.data_ptr(), n_vec, reps, blocks)
        e1.record(sb)
        time.sleep(0.003)
        step_times(2)
        ms, dom = step_times(args.steps)
        still = not sb.query()
        e1.synchronize()
        co_s = e0.elapsed_time(e1) * 1e-3
        row = {"partner": f"hbm float4 copy, {blocks} blocks", "step_ms": ms, "step_slowdown": ms / base_ms, "dominant_us": dom,
               "dominant_slowdown": dom / base_dom, "partner_outlived_steps": bool(still), "partner_TBps_alone": alone_tbs,
               "partner_TBps_whole_run_incl_tail_alone": reps * 2 * big_a.numel() * 4 / co_s * 1e-12}
        res["partners"].append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Device-resident timings of every BASELINE.json config that runs on one GPU (development tool;
the contract line is bench.py's).  Writes one JSON object with a row per config.
Usage (GPU box, repo root): python3 tools/measure_configs.py profiles/r02_configs.json"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/configs.json"
    import torch

    import _oflk
    from oflk_synth import synth_pair

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    rows = []

    def run(label, B, H, W, levels, win, iters, reps, mode="f32"):
        """mode: f32 (exact path, float32 frames), u8 (exact path, uint8 frames read by the kernels),
        fp16 (BASELINE config 5: fp16 gradients / accumulators, single-scale only), tolerant (opt-in OFLK_ARITH_TOLERANT),
        tile (single-scale with the tile kernel throughout: what every round before round 4 measured)"""
        host = [synth_pair(H, W, i) for i in range(min(B, 2))]
        prev = torch.stack([torch.from_numpy(host[b % len(host)][0]) for b in range(B)]).to(dev)
        curr = torch.stack([torch.from_numpy(host[b % len(host)][1]) for b in range(B)]).to(dev)
        u, v = torch.empty_like(prev), torch.empty_like(prev)
        if mode == "u8":
            prev, curr = prev.to(torch.uint8), curr.to(torch.uint8)   # synthetic frames are integer-valued
        plan = _oflk.Plan(0, B, H, W, levels, win, iters)
        if mode == "tolerant":
            plan.set_arithmetic(2)
        if mode == "tile":
            plan.set_kernels(1)
        if mode == "fp16":
            call = lambda *a: plan.single_scale_fp16(a[0], a[1], a[2], a[3], 255.0, a[4])  # noqa: E731
        elif mode == "u8":
            call = plan.single_scale_u8 if iters == 0 else plan.pyramidal_u8
        else:
            call = plan.single_scale if iters == 0 else plan.pyramidal
        for _ in range(3):
            call(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        plan.close()
        n = B * H * W
        # SURVEY.md section 8d byte model: 16 B/px single-scale (10 B/px with uint8 frames), 119.5 B/px for 3 levels x 3 iterations
        bytes_per_px = (10.0 if mode == "u8" else 16.0) if iters == 0 else None
        row = {"config": label, "mode": mode, "pairs": B, "shape": [H, W], "levels": levels, "window": win, "iterations": iters,
               "us_per_call": round(dt * 1e6, 1), "Mpix_per_s": round(n / dt / 1e6, 1)}
        if bytes_per_px:
            row["algorithmic_GBs"] = round(n * bytes_per_px / dt / 1e9, 1)
            row["frac_of_8TBs"] = round(n * bytes_per_px / dt / 8e12, 4)
        rows.append(row)
        print(json.dumps(row), flush=True)

    run("configs[1]: 640x480 pair, single-scale 5x5", 1, 480, 640, 1, 5, 0, 200)
    run("configs[1] batched x256", 256, 480, 640, 1, 5, 0, 20)
    run("configs[1] batched x256, tile kernel throughout (rounds 1-3)", 256, 480, 640, 1, 5, 0, 20, "tile")
    run("single-scale 5x5, 1920x1080 x32", 32, 1080, 1920, 1, 5, 0, 20)
    run("single-scale 5x5, 1920x1080 x32, tile kernel throughout", 32, 1080, 1920, 1, 5, 0, 20, "tile")
    run("configs[2]: 1920x1080 pair, 3-level pyramidal 5x5 x3", 1, 1080, 1920, 3, 5, 3, 100)
    run("configs[2] batched x32 (the batch the kernel analysis of DESIGN.md section 5 is made on)", 32, 1080, 1920, 3, 5, 3, 20)
    run("configs[2] batched x128 (bench.py workload)", 128, 1080, 1920, 3, 5, 3, 10)
    run("configs[2] batched x32, opt-in tolerant arithmetic", 32, 1080, 1920, 3, 5, 3, 20, "tolerant")
    run("configs[2] batched x128, opt-in tolerant arithmetic", 128, 1080, 1920, 3, 5, 3, 10, "tolerant")
    run("configs[3] share (8 pairs of 4K), opt-in tolerant arithmetic", 8, 2160, 3840, 3, 5, 3, 10, "tolerant")
    run("configs[3]: 3840x2160, one GPU's share of 64 pairs over 8 GPUs (8 pairs)", 8, 2160, 3840, 3, 5, 3, 10)
    run("configs[1] batched x256, uint8 frames (2 B/px of frame traffic)", 256, 480, 640, 1, 5, 0, 20, "u8")
    run("configs[2] batched x32, uint8 frames", 32, 1080, 1920, 3, 5, 3, 20, "u8")
    run("configs[4]: 7680x4320 pair, 7x7 window, fp16 gradients/accumulators (single-scale)", 1, 4320, 7680, 1, 7, 0, 20, "fp16")
    run("configs[4] geometry, exact fp32 arithmetic for comparison", 1, 4320, 7680, 1, 7, 0, 20)
    run("configs[4] geometry, exact fp32 arithmetic, tile kernel throughout (rounds 1-3)", 1, 4320, 7680, 1, 7, 0, 20, "tile")
    run("configs[4] geometry, 3-level pyramidal 7x7 x3, exact fp32", 1, 4320, 7680, 3, 7, 3, 10)
    # the RTL-bit-accurate integer mode, the RTL's own frame size
    for B, H, W, reps in ((1, 240, 320, 200), (256, 240, 320, 20), (64, 512, 1024, 20)):
        p8 = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device=dev)
        c8 = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device=dev)
        M = (H - 4) * (W - 4)
        du = torch.empty((B, M), dtype=torch.int16, device=dev)
        dv = torch.empty_like(du)
        fn = lambda: _oflk.check(_oflk.lib().oflk_rtl_flow_u8_device(p8.data_ptr(), c8.data_ptr(), B, H, W, du.data_ptr(), dv.data_ptr(), stream))  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        n = B * H * W
        row = {"config": f"RTL-bit-accurate integer mode (f3), {W}x{H} uint8 frames", "mode": "rtl_int", "pairs": B, "shape": [H, W],
               "us_per_call": round(dt * 1e6, 1), "Mpix_per_s": round(n / dt / 1e6, 1), "algorithmic_GBs": round(n * 6.0 / dt / 1e9, 1),
               "frac_of_8TBs": round(n * 6.0 / dt / 8e12, 4), "note": "2 B/px in (two uint8 frames), 4 B/px out (two int16 planes)"}
        rows.append(row)
        print(json.dumps(row), flush=True)
    Path(out_path).write_text(json.dumps({"device": torch.cuda.get_device_name(0), "inputs": "resident in HBM, synthetic",
                                          "rows": rows}, indent=1))


if __name__ == "__main__":
    main()

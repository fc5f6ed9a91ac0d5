#!/usr/bin/env python3
"""GPU golden model of the reference's single-scale RTL: what an xsim run of
``tb/tb_optical_flow_top.sv`` over ``rtl/unopt/optical_flow_top.sv`` samples, prints and writes, from the
integer kernel of liboflk (``oflk_rtl_flow_u8``: rtl/unopt/gradient_compute.sv:89-139,
window_accumulator.sv:100-189, flow_solver.sv:82-149 on the window geometry of
rtl/common/line_buffer_5x5.sv:75-151).

    python rtl_golden_model.py tb/test_frames/frame_00.mem tb/test_frames/frame_01.mem \\
        --width 320 --height 240 --output flow_field.txt

The per-element flows come from the GPU; this module adds the host-side bookkeeping of the testbench
(tb_optical_flow_top.sv:176-240, :331-360): which states the monitor loop samples and how often, the lagging
position it files each vector under, the summary it prints and the ``flow_field.txt`` it exports (the
format ``visualize_flow.py`` reads).  PARITY UNPINNED: see include/oflk.h -- no simulator output of the RTL
as committed exists to pin the model; tests hold it equal to a cycle-by-cycle execution of the modules.
There is no CPU fallback: without liboflk and a GPU this module raises.
"""
from __future__ import annotations

import argparse
import ctypes
import math
import sys
from pathlib import Path
from typing import Dict, Tuple

import numpy as np

import _oflk

# the testbench's constants (tb_optical_flow_top.sv:37-51)
GROUND_TRUTH_U, GROUND_TRUTH_V = 2.0, 0.0
EXPECTED_U_MAGNITUDE = 0.5
TEST_REGION = {"x_min": 55, "x_max": 85, "y_min": 105, "y_max": 135}
TAIL_CLOCKS = 5   # clocks of the pipeline that fall after `done`: their vectors are never sampled


def rtl_flow_states(prev: np.ndarray, curr: np.ndarray) -> Dict[str, np.ndarray]:
    """Per element k of the accumulator's gradient stream: `valid` (window_valid after ingesting k), the
    position (x, y) = flow_x / flow_y, and the S8.7 integers u, v = flow_u / flow_v.  Frames: uint8 [H, W]
    or [B, H, W]; the arrays come back [B, (H-4)(W-4)] (or without B for one pair)."""
    p, c = np.ascontiguousarray(prev), np.ascontiguousarray(curr)
    if p.dtype != np.uint8 or c.dtype != np.uint8:
        raise ValueError("the RTL takes 8-bit frames: uint8 arrays expected")
    if p.shape != c.shape or p.ndim not in (2, 3):
        raise ValueError(f"two frames (or two batches) of the same shape expected, got {p.shape} and {c.shape}")
    single = p.ndim == 2
    if single:
        p, c = p[None], c[None]
    B, H, W = p.shape
    M = int(_oflk.lib().oflk_rtl_stream_length(H, W))
    u = np.empty((B, M), np.int16)
    v = np.empty((B, M), np.int16)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    _oflk.check(_oflk.lib().oflk_rtl_flow_u8(vp(p), vp(c), B, H, W, vp(u), vp(v)))
    # the accumulator's own counters (line_buffer_5x5.sv:62-83, :136-147): its rows are W gradients long
    k = np.arange(M)
    r2, c2 = np.divmod(k, W)
    valid = (r2 >= 4) & (c2 >= 4)
    col, row = np.where(c2 == W - 1, 0, c2 + 1), np.where(c2 == W - 1, r2 + 1, r2)
    has_xy = valid & (col >= 2) & (row >= 2)
    out = {"valid": valid, "x": np.where(has_xy, col - 2, 0), "y": np.where(has_xy, row - 2, 0), "u": u, "v": v}
    if single:
        out["u"], out["v"] = u[0], v[0]
    return out


def testbench_vectors(prev: np.ndarray, curr: np.ndarray) -> np.ndarray:
    """[N, 4] (flow_x, flow_y, flow_u, flow_v) in the order the monitor loop of tb_optical_flow_top.sv:176-240
    samples them: one sample per clock in which flow_valid is high.  The accumulator holds its state while no
    gradient arrives (the four clocks at the start of every image row), so those vectors appear five times; the
    last TAIL_CLOCKS clocks of the pipeline come after `done`."""
    if prev.ndim != 2:
        raise ValueError("one frame pair")
    H, W = prev.shape
    st = rtl_flow_states(prev, curr)
    r, c = np.divmod(np.arange(H * W), W)
    n = np.nonzero((r >= 4) & (c >= 4))[0]          # stream positions whose gradient window is valid
    clock = np.arange(H * W - TAIL_CLOCKS)
    k = np.searchsorted(n, clock, side="right") - 1   # the element the accumulator ingested last
    k = k[k >= 0]
    k = k[st["valid"][k]]
    return np.stack([st["x"][k], st["y"][k], st["u"][k].astype(np.int64), st["v"][k].astype(np.int64)], axis=1)


def filed_positions(vectors: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """the position the testbench files each vector under: that of the PREVIOUS sample (it updates
    pixel_x / pixel_y after using them, :230-232); the first vector keeps its own (:185-187)"""
    if len(vectors) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    x = np.concatenate([vectors[:1, 0], vectors[:-1, 0]])
    y = np.concatenate([vectors[:1, 1], vectors[:-1, 1]])
    return x, y


def testbench_summary(vectors: np.ndarray, region: Dict[str, int] = TEST_REGION) -> Dict[str, object]:
    """the numbers of the "Results Summary" block (tb_optical_flow_top.sv:248-326)"""
    x, y = filed_positions(vectors)
    uf, vf = vectors[:, 2] / 128.0, vectors[:, 3] / 128.0     # fixed_to_float, :111-115
    inside = (x >= region["x_min"]) & (x <= region["x_max"]) & (y >= region["y_min"]) & (y <= region["y_max"])
    n = int(inside.sum())
    res: Dict[str, object] = {"valid_flow_count": int(len(vectors)), "test_region_count": n,
                              "first_vector_position": (int(vectors[0, 0]), int(vectors[0, 1])) if len(vectors) else None}
    if n:
        mu, mv = float(uf[inside].sum() / n), float(vf[inside].sum() / n)
        var_u, var_v = float((uf[inside] ** 2).sum() / n - mu * mu), float((vf[inside] ** 2).sum() / n - mv * mv)
        mag = math.sqrt(mu * mu + mv * mv)
        res.update(mean_u=mu, mean_v=mv, std_u=math.sqrt(max(var_u, 0.0)), std_v=math.sqrt(max(var_v, 0.0)),
                   error_u=mu - GROUND_TRUTH_U, error_v=mv - GROUND_TRUTH_V, flow_magnitude=mag,
                   flow_angle=math.atan2(mv, mu) * 180.0 / 3.14159,
                   passed=bool(mag >= EXPECTED_U_MAGNITUDE and abs(mv) < 0.5),
                   samples=[(int(a), int(b), float(p), float(q))
                            for a, b, p, q in zip(x[inside][::100], y[inside][::100], uf[inside][::100], vf[inside][::100])])
    return res


def print_summary(s: Dict[str, object]) -> None:
    """the testbench's $display lines (:226-228, :277-326)"""
    for a, b, p, q in s.get("samples", []):
        print("  [x=%3d, y=%3d] u=%6.3f, v=%6.3f" % (a, b, p, q))
    if not s["test_region_count"]:
        print("\n*** ERROR: No flow vectors in test region ***")
        return
    print("\n============================================\nResults Summary\n============================================")
    print(f"Total valid flow vectors: {s['valid_flow_count']}")
    print(f"Vectors in test region: {s['test_region_count']}")
    print("\nFlow Statistics (Test Region):")
    print("  Mean:         u=%6.3f, v=%6.3f" % (s["mean_u"], s["mean_v"]))
    print("  Std Dev:      u=%6.3f, v=%6.3f" % (s["std_u"], s["std_v"]))
    print("  Ground truth: u=%6.3f, v=%6.3f" % (GROUND_TRUTH_U, GROUND_TRUTH_V))
    print("  Error vs GT:  u=%6.3f, v=%6.3f" % (s["error_u"], s["error_v"]))
    print("\n============================================")
    print("Flow magnitude: %.3f pixels" % s["flow_magnitude"])
    print("Flow direction: %.1f degrees" % s["flow_angle"])
    print("*** TEST PASSED ***" if s["passed"] else "*** TEST FAILED ***")


def write_flow_field(path, vectors: np.ndarray, width: int, height: int, region: Dict[str, int] = TEST_REGION) -> None:
    """flow_field.txt as the testbench exports it (:331-360): header, then `x y u v` per sampled vector"""
    x, y = filed_positions(vectors)
    with open(path, "w") as f:
        f.write("# Optical flow field data\n# Format: x y u v\n")
        f.write(f"# Image size: {width}x{height}\n")
        f.write(f"# Test region: x[{region['x_min']}:{region['x_max']}], y[{region['y_min']}:{region['y_max']}]\n")
        for a, b, p, q in zip(x, y, vectors[:, 2] / 128.0, vectors[:, 3] / 128.0):
            f.write("%d %d %.6f %.6f\n" % (a, b, p, q))


def compare_flow_field(path, vectors: np.ndarray, max_report: int = 10) -> Dict[str, object]:
    """Pin the model against a simulator: compare a `flow_field.txt` written by an xsim run of
    tb_optical_flow_top.sv (same frames) with the model's vectors, line for line (positions as filed, u / v as the
    %.6f of the S8.7 value / 128).  Returns counts and the first differing lines."""
    rows = [ln.split() for ln in Path(path).read_text().splitlines() if ln.strip() and not ln.startswith("#")]
    x, y = filed_positions(vectors)
    mine = [("%d" % a, "%d" % b, "%.6f" % p, "%.6f" % q) for a, b, p, q in zip(x, y, vectors[:, 2] / 128.0, vectors[:, 3] / 128.0)]
    diffs = []
    for i in range(min(len(rows), len(mine))):
        if tuple(rows[i]) != mine[i]:
            diffs.append((i, tuple(rows[i]), mine[i]))
    return {"simulator_vectors": len(rows), "model_vectors": len(mine), "differing_lines": len(diffs) + abs(len(rows) - len(mine)),
            "first_differences": diffs[:max_report], "equal": len(rows) == len(mine) and not diffs}


def read_mem(path, width: int, height: int) -> np.ndarray:
    """a frame in the $readmemh format the frame buffer loads (one 2-digit hex pixel per line)"""
    vals = [int(t, 16) for t in Path(path).read_text().split() if not t.startswith("//")]
    if len(vals) != width * height:
        raise ValueError(f"{path}: {len(vals)} pixels, expected {width * height}")
    return np.array(vals, np.uint8).reshape(height, width)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="GPU golden model of the single-scale optical-flow RTL")
    ap.add_argument("frame_0", help="previous frame (.mem: $readmemh hex, or .bin: raw uint8)")
    ap.add_argument("frame_1", help="current frame")
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--output", default=None, help="write flow_field.txt here")
    ap.add_argument("--compare", default=None, metavar="FLOW_FIELD_TXT",
                    help="a flow_field.txt exported by an xsim run of tb/tb_optical_flow_top.sv on the same frames: "
                         "compare it with the model line for line (exit status 1 if they differ)")
    args = ap.parse_args(argv)

    def load(p):
        if str(p).endswith(".bin"):
            return np.fromfile(p, np.uint8).reshape(args.height, args.width)
        return read_mem(p, args.width, args.height)

    f0, f1 = load(args.frame_0), load(args.frame_1)
    vec = testbench_vectors(f0, f1)
    print_summary(testbench_summary(vec))
    if args.output:
        write_flow_field(args.output, vec, args.width, args.height)
        print(f"\nExporting {len(vec)} flow vectors to {args.output}...")
    if args.compare:
        c = compare_flow_field(args.compare, vec)
        print(f"\nsimulator: {c['simulator_vectors']} vectors, model: {c['model_vectors']}, differing lines: {c['differing_lines']}")
        for i, sim, mod in c["first_differences"]:
            print(f"  line {i}: simulator {' '.join(sim)} | model {' '.join(mod)}")
        print("MODEL PINNED: equal to the simulator's export" if c["equal"] else "MODEL DIFFERS from the simulator's export")
        return 0 if c["equal"] else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Single-scale demo CLI: counterpart of the reference's python/lucas_kanade_reference.py
(same options, same output files: raw float32 flow_u.bin / flow_v.bin, the
``x y u v`` text dump, an optional quiver PNG), with the flow computed by the HIP
kernels through the drop-in ``lucas_kanade_core`` module.  Host-side glue only.
"""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Optional

import numpy as np

from lucas_kanade_core import compute_gradients, lucas_kanade_single_scale

SCRIPT_DIR = Path(__file__).resolve().parent
PROJECT_ROOT = SCRIPT_DIR.parent
DEFAULT_FRAME_DIR = PROJECT_ROOT / "tb" / "test_frames"
DEFAULT_OUTPUT_DIR = SCRIPT_DIR / "output"


def visualize_flow(u, v, output_path: Path, scale: float = 10.0) -> None:
    """Quiver plot of every 10th flow vector (reference :22-75)."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    h, w = u.shape
    step = 10
    ys, xs = np.mgrid[step:h:step, step:w:step]
    us, vs = u[step:h:step, step:w:step], v[step:h:step, step:w:step]
    fig, ax = plt.subplots(figsize=(12, 9))
    q = ax.quiver(xs, ys, us, vs, np.hypot(us, vs), angles="xy", scale_units="xy", scale=1.0 / scale,
                  cmap="jet", width=0.003)
    ax.set_aspect("equal")
    ax.set_xlim(0, w)
    ax.set_ylim(h, 0)
    ax.set_title("Optical Flow Vectors (Single-Scale Lucas-Kanade)")
    ax.set_xlabel("X (pixels)")
    ax.set_ylabel("Y (pixels)")
    fig.colorbar(q, ax=ax, label="Flow Magnitude (pixels)")
    fig.tight_layout()
    fig.savefig(output_path, dpi=150)
    plt.close(fig)
    print(f"Flow visualization saved: {output_path}")


def export_flow_field_txt(u, v, output_path: Path, width: int, height: int,
                          test_region: Optional[dict] = None) -> None:
    """``x y u v`` per line, the format scripts/visualize_flow.py and the RTL testbench dump use
    (reference :78-103)."""
    head = ["# Optical flow field data (Python reference)", "# Format: x y u v", f"# Image size: {width}x{height}"]
    if test_region:
        head.append(f"# Test region: x[{test_region['x_min']}:{test_region['x_max']}], "
                    f"y[{test_region['y_min']}:{test_region['y_max']}]")
    with open(output_path, "w") as f:
        f.write("\n".join(head) + "\n")
        for y in range(height):
            f.write("".join(f"{x} {y} {u[y, x]:.6f} {v[y, x]:.6f}\n" for x in range(width)))
    print(f"Flow field text export: {output_path}")


def main() -> None:
    ap = argparse.ArgumentParser(description="Lucas-Kanade single-scale reference (MI355X)")
    ap.add_argument("--frame-dir", type=str, default=str(DEFAULT_FRAME_DIR),
                    help="Directory containing frame_00.bin and frame_01.bin")
    ap.add_argument("--width", type=int, default=320, help="Frame width")
    ap.add_argument("--height", type=int, default=240, help="Frame height")
    ap.add_argument("--window-size", type=int, default=5, help="Window size for Lucas-Kanade")
    ap.add_argument("--output-dir", type=str, default=str(DEFAULT_OUTPUT_DIR), help="Output directory for results")
    args = ap.parse_args()

    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    frame_dir = Path(args.frame_dir)
    shape = (args.height, args.width)
    prev = np.fromfile(frame_dir / "frame_00.bin", dtype=np.uint8).reshape(shape).astype(np.float32)
    curr = np.fromfile(frame_dir / "frame_01.bin", dtype=np.uint8).reshape(shape).astype(np.float32)
    print(f"Loaded frames: {args.width}x{args.height}")
    print(f"Window size: {args.window_size}x{args.window_size}")

    print("\nComputing gradients...")
    Ix, Iy, It = compute_gradients(prev, curr)
    print("\nGradient statistics:")
    for name, g in (("Ix", Ix), ("Iy", Iy), ("It", It)):
        print(f"  {name} range: [{np.min(g):.2f}, {np.max(g):.2f}]")

    print("Computing optical flow...")
    u, v = lucas_kanade_single_scale(prev, curr, window_size=args.window_size)

    hw = args.window_size // 2
    interior = np.s_[hw:-hw, hw:-hw]
    print("\nWindow analysis:")
    print(f"  Total possible windows: {(args.height - 2 * hw) * (args.width - 2 * hw)}")
    print(f"  Windows with non-zero flow: {np.sum(u[interior] != 0)}")

    region = np.s_[105:135, 55:85]  # the region the RTL testbench scores (tb/tb_optical_flow_top.sv)
    print("\n=== Results ===")
    print(f"Mean flow in square region: u={np.mean(u[region]):.3f}, v={np.mean(v[region]):.3f}")
    print(f"Std dev in square region:   u={np.std(u[region]):.3f}, v={np.std(v[region]):.3f}")
    print("Expected: u=2.0, v=0.0")

    u.tofile(out_dir / "flow_u.bin")
    v.tofile(out_dir / "flow_v.bin")
    print(f"\nFlow fields saved to {out_dir}")
    export_flow_field_txt(u, v, out_dir / "flow_field_python.txt", args.width, args.height,
                          {"x_min": 55, "x_max": 85, "y_min": 105, "y_max": 135})
    try:
        visualize_flow(u, v, out_dir / "flow_visualization_single_scale.png")
    except ImportError:
        print("Matplotlib not available, skipping visualization")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Four-panel diagnostic plot of an ``x y u v`` flow dump: counterpart of the reference's
``scripts/visualize_flow.py`` (parser :23-60, plot :63-247, CLI :250-321) with the same command line,
function names and panel layout.  The dump is what ``lucas_kanade_reference.py`` and the RTL testbench
write (``# Image size: WxH`` / ``# Test region: x[a:b], y[c:d]`` header, one vector per line).
Host-side plotting only; nothing here touches the GPU.

    python visualize_flow.py python/output/flow_field.txt --frame tb/test_frames/frame_00.png \\
        --output results/flow_visualization.png --ground-truth-u 2.0 --ground-truth-v 0.0
"""
from __future__ import annotations

import argparse
import re
from pathlib import Path
from typing import Dict, Tuple

import numpy as np

_SIZE = re.compile(r"Image size:\s*(\d+)\s*x\s*(\d+)")
_REGION = re.compile(r"Test region:\s*x\[(\d+):(\d+)\]\s*,\s*y\[(\d+):(\d+)\]")


def parse_flow_field(file_path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, Dict[str, int]]:
    """(x, y, u, v, metadata) of a flow dump; metadata holds width/height and test_{x,y}_{min,max}
    when the header states them (reference :23-60)."""
    meta: Dict[str, int] = {}
    with open(file_path) as f:
        for line in f:
            if not line.startswith("#"):
                break
            m = _SIZE.search(line)
            if m:
                meta["width"], meta["height"] = int(m.group(1)), int(m.group(2))
            m = _REGION.search(line)
            if m:
                meta["test_x_min"], meta["test_x_max"], meta["test_y_min"], meta["test_y_max"] = (int(g) for g in m.groups())
    data = np.atleast_2d(np.loadtxt(file_path, comments="#"))
    if data.size == 0:
        data = np.zeros((0, 4))
    return data[:, 0], data[:, 1], data[:, 2], data[:, 3], meta


def flow_grids(x, y, u, v, height: int, width: int):
    """Dense u / v / magnitude images from the vector list (pixels without a vector stay 0)."""
    uf = np.zeros((height, width))
    vf = np.zeros((height, width))
    xi, yi = x.astype(int), y.astype(int)
    ok = (xi >= 0) & (xi < width) & (yi >= 0) & (yi < height)
    uf[yi[ok], xi[ok]] = u[ok]
    vf[yi[ok], xi[ok]] = v[ok]
    return uf, vf, np.hypot(uf, vf)


def region_statistics(u, v, uf, vf, mf, meta: Dict[str, int]) -> Dict[str, float]:
    """Mean / std of u, v and |flow| over the test region's non-zero vectors (inclusive bounds, as the
    reference slices them, :113-133), or over all vectors when the header names no region."""
    keys = ("test_x_min", "test_x_max", "test_y_min", "test_y_max")
    if all(k in meta for k in keys):
        sl = (slice(meta["test_y_min"], meta["test_y_max"] + 1), slice(meta["test_x_min"], meta["test_x_max"] + 1))
        keep = mf[sl] > 0
        su, sv, sm = uf[sl][keep], vf[sl][keep], mf[sl][keep]
    else:
        su, sv, sm = u, v, np.hypot(u, v)
    stat = lambda a, f: float(f(a)) if len(a) else 0.0  # noqa: E731
    return {"mean_u": stat(su, np.mean), "mean_v": stat(sv, np.mean), "std_u": stat(su, np.std), "std_v": stat(sv, np.std),
            "mean_mag": stat(sm, np.mean), "std_mag": stat(sm, np.std), "num_vectors": int(len(su))}


def _load_frame(frame_path: str, meta: Dict[str, int]) -> np.ndarray:
    p = Path(frame_path)
    if p.suffix.lower() == ".bin":   # raw 8-bit frame (generate_test_suite.py:259-261); size from the dump's header
        return np.fromfile(p, np.uint8).reshape(meta["height"], meta["width"])
    from PIL import Image

    return np.array(Image.open(p).convert("L"))


def create_diagnostic_plot(frame_path: str, flow_file: str, output_path: str, ground_truth_u: float = 2.0,
                           ground_truth_v: float = 0.0, stride: int = 10, scale: float = 20.0) -> Dict[str, float]:
    """Quiver overlay, magnitude heat map, component histogram with statistics, error magnitude against
    the constant ground truth (reference :63-247).  Returns the statistics it prints into the plot."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    from matplotlib.patches import Rectangle

    x, y, u, v, meta = parse_flow_field(flow_file)
    frame = _load_frame(frame_path, meta)
    height, width = frame.shape
    uf, vf, mf = flow_grids(x, y, u, v, height, width)
    err = np.hypot(uf - ground_truth_u, vf - ground_truth_v)
    st = region_statistics(u, v, uf, vf, mf, meta)
    has_region = "test_x_min" in meta and "test_y_min" in meta

    def outline(ax, colour):
        if has_region:
            ax.add_patch(Rectangle((meta["test_x_min"], meta["test_y_min"]), meta["test_x_max"] - meta["test_x_min"],
                                   meta["test_y_max"] - meta["test_y_min"], linewidth=2, edgecolor=colour,
                                   facecolor="none", linestyle="--"))

    fig = plt.figure(figsize=(16, 12), dpi=150)
    grid = fig.add_gridspec(2, 2, hspace=0.3, wspace=0.3)
    extent = (0, width, height, 0)

    ax = fig.add_subplot(grid[0, 0])
    ax.imshow(frame, cmap="gray", extent=extent)
    pick = np.arange(0, len(x), max(int(stride), 1))
    ax.quiver(x[pick], y[pick], u[pick], -v[pick], np.hypot(u, v)[pick], cmap="jet", scale=scale, width=0.003)
    outline(ax, "lime")
    ax.set_title("Optical Flow Field (Quiver Plot)", fontsize=12, pad=10)
    ax.set_xlim(0, width), ax.set_ylim(height, 0)

    ax = fig.add_subplot(grid[0, 1])
    im = ax.imshow(mf, cmap="hot", extent=extent)
    outline(ax, "cyan")
    ax.set_title("Flow Magnitude Heatmap", fontsize=12, pad=10)
    fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04).set_label("Magnitude (pixels)", rotation=270, labelpad=15)

    ax = fig.add_subplot(grid[1, 0])
    if has_region:
        sl = (slice(meta["test_y_min"], meta["test_y_max"] + 1), slice(meta["test_x_min"], meta["test_x_max"] + 1))
        keep = mf[sl] > 0
        hu, hv = uf[sl][keep], vf[sl][keep]
    else:
        hu, hv = u, v
    ax.hist(hu, bins=50, alpha=0.6, color="blue", label="u (horizontal)", edgecolor="black")
    ax.hist(hv, bins=50, alpha=0.6, color="red", label="v (vertical)", edgecolor="black")
    ax.axvline(ground_truth_u, color="blue", linestyle="--", linewidth=1)
    ax.axvline(ground_truth_v, color="red", linestyle="--", linewidth=1)
    ax.set_xlabel("Flow (pixels)"), ax.set_ylabel("Frequency")
    ax.set_title("Flow Component Distribution", fontsize=12, pad=10)
    ax.legend(loc="upper right"), ax.grid(True, alpha=0.3)
    ax.text(0.02, 0.98,
            f"Flow Statistics:\nMean: u={st['mean_u']:.3f}, v={st['mean_v']:.3f}\nStd:  u={st['std_u']:.3f}, v={st['std_v']:.3f}\n"
            f"Magnitude: {st['mean_mag']:.3f} ± {st['std_mag']:.3f}\nTotal vectors: {len(u)}\n\n"
            f"Test Region ({st['num_vectors']} vectors):\nMean: u={st['mean_u']:.3f}, v={st['mean_v']:.3f}\n"
            f"Magnitude: {st['mean_mag']:.3f} ± {st['std_mag']:.3f}\n"
            f"Error vs GT: u={st['mean_u'] - ground_truth_u:.3f}, v={st['mean_v'] - ground_truth_v:.3f}",
            transform=ax.transAxes, fontsize=9, verticalalignment="top", family="monospace",
            bbox=dict(boxstyle="round,pad=0.5", facecolor="wheat", alpha=0.8))

    ax = fig.add_subplot(grid[1, 1])
    im = ax.imshow(err, cmap="viridis", extent=extent, vmin=0, vmax=float(err.max()) if err.size else 1.0)
    outline(ax, "cyan")
    ax.set_title("Error Magnitude vs Ground Truth", fontsize=12, pad=10)
    ax.set_xlabel("X (pixels)"), ax.set_ylabel("Y (pixels)"), ax.set_aspect("equal")
    fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04).set_label("Error (pixels)", rotation=270, labelpad=15)

    Path(output_path).parent.mkdir(parents=True, exist_ok=True)
    fig.savefig(output_path, dpi=150, bbox_inches="tight")
    plt.close(fig)
    return st


def main() -> None:
    ap = argparse.ArgumentParser(description="Visualize optical flow results")
    ap.add_argument("flow_file", help="Flow field file to visualize")
    ap.add_argument("--frame", default="tb/test_frames/frame_00.png", help="Frame to overlay flow on (PNG, or raw .bin)")
    ap.add_argument("--output", default="results/flow_visualization.png", help="Output visualization file")
    ap.add_argument("--ground-truth-u", type=float, default=2.0, help="Ground truth horizontal flow (pixels)")
    ap.add_argument("--ground-truth-v", type=float, default=0.0, help="Ground truth vertical flow (pixels)")
    ap.add_argument("--stride", type=int, default=10, help="Arrow subsampling stride")
    ap.add_argument("--scale", type=float, default=20.0, help="Arrow scale factor")
    args = ap.parse_args()
    if not Path(args.flow_file).exists():
        print(f"ERROR: Flow file not found: {args.flow_file}")
        return
    if not Path(args.frame).exists():
        print(f"ERROR: Frame file not found: {args.frame}")
        print("Run: python scripts/convert_frames.py")
        return
    print("Generating 4-panel diagnostic visualization...")
    create_diagnostic_plot(args.frame, args.flow_file, args.output, args.ground_truth_u, args.ground_truth_v,
                           args.stride, args.scale)
    print(f"Generated: {args.output}")
    print("\nVisualization complete!")


if __name__ == "__main__":
    main()

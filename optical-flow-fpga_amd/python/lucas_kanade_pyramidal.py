"""Drop-in for the reference's ``python/lucas_kanade_pyramidal.py``: same module
name, function names, signatures and defaults; the arithmetic runs as HIP kernels
on an MI355X through liboflk's C ABI (no CPU fallback).

``lucas_kanade_pyramidal`` is one C call (pyramids, warps, fused LK iterations,
residual reduction and the data-dependent early exit all stay on the device);
the progress lines the reference prints (lucas_kanade_pyramidal.py:172-222) are
reproduced afterwards from the residual log the call returns.

The reference also dumps python/output/pyramid_level_{l}.png from inside the
function (:226), on every call.  Here that side effect is opt-in and best-effort:
with OFLK_DUMP_LEVELS=1 every level's final flow is read back from the device
after the call and handed to ``visualize_pyramid_level`` (skipped silently when
matplotlib is missing); without it nothing leaves the device but the result.

Environment switches (host-side behaviour only, never the arithmetic):
  OFLK_QUIET=1         suppress the progress lines
  OFLK_DUMP_LEVELS=1   write python/output/pyramid_level_{l}.png like the reference
                       (OFLK_DUMP_DIR overrides the directory)
"""
from __future__ import annotations

import argparse
import ctypes
import os
from pathlib import Path
from typing import List, Tuple

import numpy as np
import numpy.typing as npt

import _oflk
from lucas_kanade_core import lucas_kanade_single_scale

SCRIPT_DIR = Path(__file__).resolve().parent
PROJECT_ROOT = SCRIPT_DIR.parent
DEFAULT_FRAME_DIR = PROJECT_ROOT / "tb" / "test_frames"

_i32p = ctypes.POINTER(ctypes.c_int)
_f32p = ctypes.POINTER(ctypes.c_float)


def _say(msg: str) -> None:
    if os.environ.get("OFLK_QUIET", "0") != "1":
        print(msg)


def pyramid_level_shapes(shape: Tuple[int, int], num_levels: int, scale_factor: float = 0.5) -> List[Tuple[int, int]]:
    """(H, W) per level, coarse to fine: int(H * scale_factor) per step (reference :51-52)."""
    dims = (ctypes.c_int * (2 * num_levels))()
    _oflk.check(_oflk.lib().oflk_pyramid_level_dims(int(shape[0]), int(shape[1]), int(num_levels),
                                                    float(scale_factor), dims))
    return [(dims[2 * l], dims[2 * l + 1]) for l in range(num_levels)]


def build_gaussian_pyramid(
    image: npt.NDArray[np.float32], num_levels: int, scale_factor: float = 0.5
) -> List[npt.NDArray[np.float32]]:
    """Gaussian pyramid, list from coarse (smallest) to fine (original).

    Replaces reference lucas_kanade_pyramidal.py:23-63 (oflk_build_pyramid):
    gaussian_filter(sigma = 1/scale_factor) then bilinear sampling on a linspace grid.
    """
    img = _oflk.as_f32(image)
    H, W = img.shape
    if num_levels < 1:
        return []  # range(0) in the reference: nothing is appended
    shapes = pyramid_level_shapes((H, W), num_levels, scale_factor)
    levels = [np.empty(s, np.float32) for s in shapes]
    arr = (_f32p * num_levels)(*[_oflk.ptr(a) for a in levels])
    # The Gaussian weights as SciPy forms them (scipy/ndimage/_filters.py _gaussian_kernel1d: NumPy's exp, normalised by
    # the kernel's sum), handed to the library: its own fallback for sigma != 2 is libm's exp, which differs from NumPy's
    # in the last bit for some arguments.  With the caller's weights every scale factor gives the reference's values.
    sigma = 1.0 / float(scale_factor)   # :46
    radius = int(4.0 * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    w = np.ascontiguousarray(phi[radius:], np.float64)
    _oflk.check(_oflk.lib().oflk_build_pyramid_w(_oflk.ptr(img), H, W, int(num_levels), float(scale_factor),
                                                 w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), radius, arr))
    return levels


def warp_image(
    image: npt.NDArray[np.float32],
    flow_u: npt.NDArray[np.float32],
    flow_v: npt.NDArray[np.float32],
) -> npt.NDArray[np.float32]:
    """Bilinear backward warp: out[y, x] = image(y + v[y, x], x + u[y, x]), 0 outside.

    Replaces reference lucas_kanade_pyramidal.py:66-97 (oflk_warp).
    """
    img, u, v = _oflk.as_f32(image), _oflk.as_f32(flow_u), _oflk.as_f32(flow_v)
    H, W = _oflk.same_shape(img, u, v)
    out = np.empty((H, W), np.float32)
    _oflk.check(_oflk.lib().oflk_warp(_oflk.ptr(img), _oflk.ptr(u), _oflk.ptr(v), H, W, _oflk.ptr(out)))
    return out


def upsample_flow(
    flow_u: npt.NDArray[np.float32],
    flow_v: npt.NDArray[np.float32],
    target_shape: Tuple[int, int],
) -> Tuple[npt.NDArray[np.float32], npt.NDArray[np.float32]]:
    """Bilinear flow upsampling with magnitudes scaled by the resolution ratio.

    Replaces reference lucas_kanade_pyramidal.py:100-138 (oflk_upsample_flow).
    """
    u, v = _oflk.as_f32(flow_u), _oflk.as_f32(flow_v)
    Hc, Wc = _oflk.same_shape(u, v)
    Ht, Wt = int(target_shape[0]), int(target_shape[1])
    uo = np.empty((Ht, Wt), np.float32)
    vo = np.empty((Ht, Wt), np.float32)
    _oflk.check(_oflk.lib().oflk_upsample_flow(_oflk.ptr(u), _oflk.ptr(v), Hc, Wc, Ht, Wt,
                                               _oflk.ptr(uo), _oflk.ptr(vo)))
    return uo, vo


def lucas_kanade_pyramidal_with_log(
    frame_prev, frame_curr, num_levels: int = 3, window_size: int = 5, num_iterations: int = 3
):
    """Like lucas_kanade_pyramidal but silent; returns (u, v, residual_log, iters_run).

    residual_log[l, k] = (mean|du|, mean|dv|) of iteration k at level l (coarse first);
    iters_run[l] = iterations executed at level l.
    """
    log = np.zeros((max(num_levels, 1), max(num_iterations, 1), 2), np.float32)
    runs = np.zeros(max(num_levels, 1), np.int32)
    if _oflk.both_u8(frame_prev, frame_curr):
        # raw 8-bit frames: converted on the device (same values as .astype(np.float32) first)
        p, c = np.ascontiguousarray(frame_prev), np.ascontiguousarray(frame_curr)
        H, W = _oflk.same_shape(p, c)
        u = np.empty((H, W), np.float32)
        v = np.empty((H, W), np.float32)
        _oflk.check(_oflk.lib().oflk_pyramidal_u8(p.ctypes.data, c.ctypes.data, 1, H, W, int(num_levels),
                                                  int(window_size), int(num_iterations), _oflk.ptr(u),
                                                  _oflk.ptr(v), _oflk.ptr(log), runs.ctypes.data_as(_i32p)))
        return u, v, log, runs
    p, c = _oflk.as_f32(frame_prev), _oflk.as_f32(frame_curr)
    H, W = _oflk.same_shape(p, c)
    u = np.empty((H, W), np.float32)
    v = np.empty((H, W), np.float32)
    _oflk.check(_oflk.lib().oflk_pyramidal(_oflk.ptr(p), _oflk.ptr(c), H, W, int(num_levels),
                                           int(window_size), int(num_iterations), _oflk.ptr(u),
                                           _oflk.ptr(v), _oflk.ptr(log), runs.ctypes.data_as(_i32p)))
    return u, v, log, runs


def lucas_kanade_pyramidal(
    frame_prev: npt.NDArray[np.float32],
    frame_curr: npt.NDArray[np.float32],
    num_levels: int = 3,
    window_size: int = 5,
    num_iterations: int = 3,
) -> Tuple[npt.NDArray[np.float32], npt.NDArray[np.float32]]:
    """Coarse-to-fine pyramidal Lucas-Kanade flow (u, v) at the input resolution.

    Replaces reference lucas_kanade_pyramidal.py:141-228 (oflk_pyramidal).
    """
    u, v, log, runs = lucas_kanade_pyramidal_with_log(frame_prev, frame_curr, num_levels,
                                                      window_size, num_iterations)

    shapes = pyramid_level_shapes(np.shape(frame_prev), num_levels)
    H, W = int(np.shape(frame_prev)[0]), int(np.shape(frame_prev)[1])
    key = (1, H, W, int(num_levels), int(window_size), int(num_iterations))
    # exit decisions taken too close to the 0.01 threshold were redone by the library in NumPy's own summation order
    # (oflk_plan_resolve_uncertain, see oflk.h): the result is the reference's either way; say so when it happened
    resolved = int(_oflk.lib().oflk_last_resolved())
    _say(f"Building {num_levels}-level Gaussian pyramids...")
    _say("Pyramid levels:")
    for i, (h, w) in enumerate(shapes):
        _say(f"  Level {i}: {w}x{h} pixels")
    for level in range(num_levels):
        _say(f"\nProcessing pyramid level {level}/{num_levels-1}...")
        if level > 0:
            _say(f"  Upsampled flow to {shapes[level][1]}x{shapes[level][0]}")
        for it in range(int(runs[level])):
            mu, mv = log[level, it]
            _say(f"  Iteration {it+1}/{num_iterations}: mean residual = ({mu:.4f}, {mv:.4f})")
            if mu < 0.01 and mv < 0.01:
                _say(f"  Converged after {it+1} iterations")
    if resolved:
        _say("  note: an exit decision of this pair lay within the summation error of the 0.01 threshold and was "
             "re-evaluated in NumPy's summation order")
    if os.environ.get("OFLK_DUMP_LEVELS", "0") == "1":
        _dump_levels(key, shapes, u, v)
    return u, v


def _dump_levels(key, shapes, u, v) -> None:
    """The reference's per-level PNG side effect (:226), after the call, best-effort."""
    num_levels = key[3]
    out_dir = os.environ.get("OFLK_DUMP_DIR", "python/output")
    try:
        for level in range(num_levels):
            if level == num_levels - 1:
                lu, lv = u, v
            else:
                lu = np.empty(shapes[level], np.float32)
                lv = np.empty(shapes[level], np.float32)
                _oflk.check(_oflk.lib().oflk_pyramidal_last_level_flow(*key, level, 0, _oflk.ptr(lu), _oflk.ptr(lv)))
            visualize_pyramid_level(lu, lv, level, num_levels, out_dir)
    except ImportError:
        pass  # no matplotlib on this box


# ---------------------------------------------------------------------------
# host-side plotting and CLI (names kept for drop-in use; not GPU work)
# ---------------------------------------------------------------------------
def _quiver_panel(ax, u, v, title: str, step: int, scale: float) -> None:
    h, w = u.shape
    ys, xs = np.mgrid[step:h:step, step:w:step]
    us, vs = u[step:h:step, step:w:step], v[step:h:step, step:w:step]
    ax.quiver(xs, ys, us, vs, np.hypot(us, vs), angles="xy", scale_units="xy",
              scale=1.0 / scale, cmap="jet", width=0.003)
    ax.set_aspect("equal")
    ax.set_xlim(0, w)
    ax.set_ylim(h, 0)
    ax.set_title(title)
    ax.set_xlabel("X (pixels)")
    ax.set_ylabel("Y (pixels)")


def visualize_flow_comparison(flow_u_single, flow_v_single, flow_u_pyr, flow_v_pyr,
                              output_path: Path, scale: float = 1.0) -> None:
    """Side-by-side quiver plots, single-scale vs pyramidal (reference :231-310)."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    fig, axes = plt.subplots(1, 2, figsize=(20, 9))
    _quiver_panel(axes[0], flow_u_single, flow_v_single, "Single-Scale Lucas-Kanade", 10, scale)
    _quiver_panel(axes[1], flow_u_pyr, flow_v_pyr, "Pyramidal Lucas-Kanade", 10, scale)
    fig.tight_layout()
    fig.savefig(output_path, dpi=100)
    plt.close(fig)
    print(f"Comparison visualization saved: {output_path}")


def visualize_pyramid_level(flow_u, flow_v, level: int, num_levels: int = 3,
                            output_dir: str = "python/output") -> None:
    """U / V / magnitude images of one level's flow (reference :313-351)."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    from matplotlib.colors import Normalize

    os.makedirs(output_dir, exist_ok=True)
    panels = [
        (flow_u, "RdBu_r", Normalize(vmin=-20, vmax=20), f"Level {level}: U (horizontal)"),
        (flow_v, "RdBu_r", Normalize(vmin=-20, vmax=20), f"Level {level}: V (vertical)"),
        (np.sqrt(flow_u**2 + flow_v**2), "viridis", Normalize(vmin=0, vmax=20), f"Level {level}: Magnitude"),
    ]
    fig, axes = plt.subplots(1, 3, figsize=(15, 4))
    for ax, (img, cmap, norm, title) in zip(axes, panels):
        im = ax.imshow(img, cmap=cmap, norm=norm)
        ax.set_title(title)
        ax.axis("off")
        fig.colorbar(im, ax=ax, label="pixels")
    fig.tight_layout()
    fig.savefig(f"{output_dir}/pyramid_level_{level}.png", dpi=100, bbox_inches="tight")
    plt.close(fig)


def _load_pair(frame_dir: Path, height: int, width: int):
    prev = np.fromfile(frame_dir / "frame_00.bin", dtype=np.uint8).reshape(height, width)
    curr = np.fromfile(frame_dir / "frame_01.bin", dtype=np.uint8).reshape(height, width)
    return prev.astype(np.float32), curr.astype(np.float32)


def main() -> None:
    """CLI with the reference's options (reference :354-475)."""
    ap = argparse.ArgumentParser(description="Pyramidal Lucas-Kanade optical flow (MI355X)")
    ap.add_argument("--frame-dir", type=str, default=str(DEFAULT_FRAME_DIR),
                    help="Directory containing frame_00.bin and frame_01.bin")
    ap.add_argument("--width", type=int, default=320, help="Frame width")
    ap.add_argument("--height", type=int, default=240, help="Frame height")
    ap.add_argument("--num-levels", type=int, default=3, help="Number of pyramid levels")
    ap.add_argument("--window-size", type=int, default=5, help="Window size")
    ap.add_argument("--num-iterations", type=int, default=3, help="Iterations per pyramid level")
    ap.add_argument("--output-dir", type=str, default="python/output", help="Output directory")
    ap.add_argument("--compare", action="store_true", help="Compare with single-scale implementation")
    args = ap.parse_args()

    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    frame_prev, frame_curr = _load_pair(Path(args.frame_dir), args.height, args.width)

    bar = "=" * 60
    print(bar + "\nPyramidal Lucas-Kanade Optical Flow\n" + bar)
    print(f"Loaded frames: {args.width}x{args.height}")
    print(f"Pyramid levels: {args.num_levels}")
    print(f"Window size: {args.window_size}x{args.window_size}")
    print(f"Iterations per level: {args.num_iterations}")
    print("\n" + bar + "\nRunning Pyramidal Lucas-Kanade...\n" + bar)
    u_pyr, v_pyr = lucas_kanade_pyramidal(frame_prev, frame_curr, num_levels=args.num_levels,
                                          window_size=args.window_size,
                                          num_iterations=args.num_iterations)

    region = np.s_[105:135, 55:85]
    u_mean, v_mean = np.mean(u_pyr[region]), np.mean(v_pyr[region])
    print("\n" + bar + "\nPyramidal Results\n" + bar)
    print(f"Mean flow in test region: u={u_mean:.3f}, v={v_mean:.3f}")
    print(f"Std dev in test region:   u={np.std(u_pyr[region]):.3f}, v={np.std(v_pyr[region]):.3f}")
    u_pyr.tofile(out_dir / "flow_u_pyramidal.bin")
    v_pyr.tofile(out_dir / "flow_v_pyramidal.bin")
    print(f"\nPyramidal flow fields saved to {out_dir}")

    if args.compare:
        print("\n" + bar + "\nRunning Single-Scale for Comparison...\n" + bar)
        u_s, v_s = lucas_kanade_single_scale(frame_prev, frame_curr, window_size=args.window_size)
        us_mean, vs_mean = np.mean(u_s[region]), np.mean(v_s[region])
        print("\n" + bar + "\nComparison\n" + bar)
        print(f"Single-scale: u={us_mean:.3f}, v={vs_mean:.3f}")
        print(f"Pyramidal:    u={u_mean:.3f}, v={v_mean:.3f}")
        print(f"Difference:   u={abs(u_mean - us_mean):.3f}, v={abs(v_mean - vs_mean):.3f}")
        try:
            visualize_flow_comparison(u_s, v_s, u_pyr, v_pyr, out_dir / "flow_comparison.png")
        except ImportError:
            print("Matplotlib not available, skipping visualization")
    print("\n" + bar + "\nComplete!\n" + bar)


if __name__ == "__main__":
    main()

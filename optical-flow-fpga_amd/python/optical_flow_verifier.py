#!/usr/bin/env python3
"""Verification driver: counterpart of the reference's python/optical_flow_verifier.py.

Same CLI flags, same function names, same report formats (markdown table,
results JSON, +-threshold regression check against a baseline JSON), so the
reference's CI recipe (generate_test_suite.py, then
``optical_flow_verifier.py --compare-baseline --no-visualizations
--regression-threshold 10.0``; .github/workflows/verify_optical_flow.yml:43-51)
drives the MI355X build unchanged.  The two flow functions it calls are this
directory's drop-in ``lucas_kanade_core`` / ``lucas_kanade_pyramidal`` (HIP).

Host-side harness code; the only GPU work is inside the two imported functions.
"""
from __future__ import annotations

import argparse
import json
import sys
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, List, Tuple

import numpy as np
import numpy.typing as npt
import yaml

from flow_metrics import compute_all_metrics, compute_all_metrics_gpu
from lucas_kanade_core import lucas_kanade_single_scale
from lucas_kanade_pyramidal import lucas_kanade_pyramidal

Flow = Tuple[npt.NDArray[np.float32], npt.NDArray[np.float32]]
REGRESSION_METRICS = ("mae_u", "mae_v", "epe")  # what the reference gates on (:608)


# ---- loading -----------------------------------------------------------------
def load_config(config_path: Path) -> Dict[str, Any]:
    with open(config_path, "r") as f:
        cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict):
        raise ValueError(f"{config_path}: expected a mapping")
    return cfg


def load_test_suite_index(suite_dir: Path) -> Dict[str, Any]:
    with open(Path(suite_dir) / "suite_index.json", "r") as f:
        return json.load(f)


def load_test_pattern(pattern_dir: Path) -> Dict[str, Any]:
    """frame_00.bin / frame_01.bin (raw uint8) -> float32 [H, W], plus metadata.json (reference :43-71)."""
    pattern_dir = Path(pattern_dir)
    with open(pattern_dir / "metadata.json", "r") as f:
        meta = json.load(f)
    w, h = meta["resolution"]["width"], meta["resolution"]["height"]
    frames = [np.fromfile(pattern_dir / n, dtype=np.uint8).reshape(h, w).astype(np.float32)
              for n in ("frame_00.bin", "frame_01.bin")]
    return {"frame_prev": frames[0], "frame_curr": frames[1], "metadata": meta}


def get_thresholds_for_pattern(pattern_name: str, config: Dict[str, Any]) -> Tuple[float, float]:
    for category, names in config["pattern_categories"].items():
        if pattern_name in names:
            t = config["thresholds"][category]
            return t["mae_pass"], t["mae_warning"]
    print(f"Warning: Unknown pattern '{pattern_name}', using translation thresholds")
    t = config["thresholds"]["translation"]
    return t["mae_pass"], t["mae_warning"]


# ---- scoring -----------------------------------------------------------------
def get_test_region_mask(shape: Tuple[int, int], pattern_type: str, center_crop_size: int) -> npt.NDArray[np.bool_]:
    """Translation patterns: whole frame minus a 10 px border; rotation / zoom /
    combined: the centred crop only (reference :96-138)."""
    h, w = shape
    mask = np.zeros((h, w), dtype=bool)
    if any(tag in pattern_type for tag in ("rotate", "zoom", "translate_rotate")):
        half = center_crop_size // 2
        cy, cx = h // 2, w // 2
        mask[cy - half:cy + half, cx - half:cx + half] = True
    else:
        mask[10:-10, 10:-10] = True
    return mask


def run_single_scale_lk(frame_prev, frame_curr, window_size: int) -> Flow:
    return lucas_kanade_single_scale(frame_prev, frame_curr, window_size)


def run_pyramidal_lk(frame_prev, frame_curr, pyramid_config: Dict[str, Any]) -> Flow:
    return lucas_kanade_pyramidal(frame_prev, frame_curr, num_levels=pyramid_config["levels"],
                                  window_size=pyramid_config["window_size"],
                                  num_iterations=pyramid_config["iterations"])


def classify_result(mae_u: float, mae_v: float, pattern_type: str, config: Dict[str, Any]) -> str:
    ok, warn = get_thresholds_for_pattern(pattern_type, config)
    worst = max(mae_u, mae_v)
    return "Pass" if worst <= ok else ("Warning" if worst <= warn else "Fail")


def verify_pattern(pattern_name: str, pattern_data: Dict[str, Any], config: Dict[str, Any],
                   pyramid_config_name: str = "default", verbose: bool = True,
                   device_metrics: bool = False) -> Dict[str, Any]:
    """Both LK variants on one pattern, metrics vs the constant ground truth (reference :211-312).
    device_metrics=True reduces the masked metrics on the GPU (oflk_flow_metrics) instead of in NumPy;
    the numbers agree to ~1e-6 relative, so the default keeps the reference's own arithmetic."""
    say = print if verbose else (lambda *a, **k: None)
    prev, curr, meta = pattern_data["frame_prev"], pattern_data["frame_curr"], pattern_data["metadata"]
    motion = meta["motion_parameters"]
    u_true, v_true = motion["dx"], motion["dy"]
    say(f"\n{'=' * 60}\nTesting: {pattern_name}\n{'=' * 60}")
    say(f"Ground truth: u={u_true:.1f}, v={v_true:.1f} pixels\nDescription: {motion['description']}")
    mask = get_test_region_mask(prev.shape, pattern_name, config["test_region"]["center_crop"])
    say(f"Test region: {int(mask.sum())} pixels")

    out: Dict[str, Any] = {"pattern_name": pattern_name, "ground_truth": {"u": u_true, "v": v_true},
                           "num_test_pixels": int(mask.sum())}
    runs = [
        ("single_scale", "single-scale Lucas-Kanade",
         lambda: run_single_scale_lk(prev, curr, config["pyramids"]["default"]["window_size"])),
        ("pyramidal", f"pyramidal Lucas-Kanade ({pyramid_config_name})",
         lambda: run_pyramidal_lk(prev, curr, config["pyramids"][pyramid_config_name])),
    ]
    for key, label, fn in runs:
        say(f"\nRunning {label}...")
        u, v = fn()
        m = (compute_all_metrics_gpu if device_metrics else compute_all_metrics)(u, v, u_true, v_true, mask)
        say(f"  MAE: u={m['mae_u']:.3f}, v={m['mae_v']:.3f}\n  RMSE: {m['rmse']:.3f}\n  EPE: {m['epe']:.3f}\n"
            f"  AAE: {m['aae']:.2f}°")
        out[key] = {"metrics": m, "status": classify_result(m["mae_u"], m["mae_v"], pattern_name, config)}
    out["pyramidal"]["config"] = pyramid_config_name
    say(f"\nSingle-scale status: {out['single_scale']['status']}\nPyramidal status: {out['pyramidal']['status']}")
    return out


# ---- reports -----------------------------------------------------------------
_TABLE_HEAD = ["| Pattern | Ground Truth | MAE (u) | MAE (v) | RMSE | EPE | AAE | Status |",
               "|---------|--------------|---------|---------|------|-----|-----|--------|"]


def _table_rows(results: List[Dict[str, Any]], key: str) -> List[str]:
    rows = []
    for r in results:
        gt, m = r["ground_truth"], r[key]["metrics"]
        rows.append(f"| {r['pattern_name']:20s} | ({gt['u']:4.1f}, {gt['v']:4.1f}) | {m['mae_u']:5.3f} | "
                    f"{m['mae_v']:5.3f} | {m['rmse']:5.3f} | {m['epe']:5.3f} | {m['aae']:5.2f}° | {r[key]['status']} |")
    return rows


def generate_markdown_table(results: List[Dict[str, Any]]) -> str:
    """Same layout as the reference's verification_results.md (reference :320-371)."""
    lines = ["# Optical Flow Verification Results\n", "## Single-Scale Lucas-Kanade\n", *_TABLE_HEAD,
             *_table_rows(results, "single_scale"), "\n## Pyramidal Lucas-Kanade\n", *_TABLE_HEAD,
             *_table_rows(results, "pyramidal"), "\n## Metrics Legend\n",
             "- **MAE**: Mean Absolute Error (pixels)", "- **RMSE**: Root Mean Square Error (pixels)",
             "- **EPE**: Average Endpoint Error (pixels)", "- **AAE**: Average Angular Error (degrees)",
             "- **Pass**: MAE within expected threshold", "- **Warning**: MAE slightly elevated but acceptable",
             "- **Fail**: MAE exceeds threshold (expected for extreme motion)"]
    return "\n".join(lines)


def _results_document(results: List[Dict[str, Any]], stamp: str) -> Dict[str, Any]:
    return {"version": "1.0", "timestamp": stamp, "patterns": {r["pattern_name"]: r for r in results}}


def save_results_json(results: List[Dict[str, Any]], output_path: Path) -> None:
    with open(output_path, "w") as f:
        json.dump(_results_document(results, datetime.now(timezone.utc).isoformat()), f, indent=2)
    print(f"\nResults saved: {output_path}")


# ---- plots (showcase patterns; optional) ---------------------------------------
def visualize_flow_field(u, v, title: str, output_path: Path, subsample_step: int = 10, scale: float = 1.0) -> None:
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    h, w = u.shape
    s = subsample_step
    ys, xs = np.mgrid[s:h:s, s:w:s]
    us, vs = u[s:h:s, s:w:s], v[s:h:s, s:w:s]
    fig, ax = plt.subplots(figsize=(12, 9))
    q = ax.quiver(xs, ys, us, vs, np.hypot(us, vs), angles="xy", scale_units="xy", scale=1.0 / scale,
                  cmap="jet", width=0.003)
    ax.set_aspect("equal")
    ax.set_xlim(0, w)
    ax.set_ylim(h, 0)
    ax.set_title(title)
    ax.set_xlabel("X (pixels)")
    ax.set_ylabel("Y (pixels)")
    fig.colorbar(q, ax=ax, label="Flow Magnitude (pixels)")
    fig.tight_layout()
    fig.savefig(output_path, dpi=150, bbox_inches="tight")
    plt.close(fig)


def visualize_error_heatmap(u_pred, v_pred, u_true: float, v_true: float, title: str, output_path: Path,
                            vmax: float = 5.0) -> None:
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    err = np.hypot(u_pred - u_true, v_pred - v_true)
    fig, ax = plt.subplots(figsize=(12, 9))
    im = ax.imshow(err, cmap="hot", vmin=0, vmax=vmax, interpolation="nearest")
    ax.set_title(title)
    ax.set_xlabel("X (pixels)")
    ax.set_ylabel("Y (pixels)")
    fig.colorbar(im, ax=ax, label="Error Magnitude (pixels)")
    fig.tight_layout()
    fig.savefig(output_path, dpi=150, bbox_inches="tight")
    plt.close(fig)


def generate_visualizations(pattern_name: str, pattern_data: Dict[str, Any], u_single, v_single, u_pyr, v_pyr,
                            output_dir: Path, config: Dict[str, Any]) -> None:
    print(f"  Generating visualizations for {pattern_name}...")
    pdir = Path(output_dir) / pattern_name
    pdir.mkdir(parents=True, exist_ok=True)
    motion = pattern_data["metadata"]["motion_parameters"]
    viz = config["visualization"]
    step, scale = viz["quiver"]["subsample_step"], viz["quiver"]["scale"]
    vmax = viz["colormap"]["error_range"][1]
    for tag, label, (u, v) in (("single", "Single-Scale", (u_single, v_single)),
                               ("pyramidal", "Pyramidal", (u_pyr, v_pyr))):
        visualize_flow_field(u, v, f"{pattern_name} - {label} L-K Flow Field", pdir / f"flow_{tag}.png", step, scale)
        visualize_error_heatmap(u, v, motion["dx"], motion["dy"], f"{pattern_name} - {label} Error",
                                pdir / f"error_{tag}.png", vmax=vmax)
    print(f"  Saved visualizations to {pdir}")


# ---- regression against a baseline ---------------------------------------------
def load_baseline(baseline_path: Path) -> Dict[str, Any]:
    baseline_path = Path(baseline_path)
    if not baseline_path.exists():
        print(f"Warning: Baseline file not found: {baseline_path}")
        return {}
    with open(baseline_path, "r") as f:
        data = json.load(f)
    if not isinstance(data, dict):
        print(f"Warning: Baseline file has invalid format: {baseline_path}")
        return {}
    return data


def compare_metrics(current: Dict[str, float], baseline: Dict[str, float], threshold_percent: float = 10.0
                    ) -> Dict[str, Any]:
    """Relative change of mae_u, mae_v, epe; a (near-)zero baseline only flags a
    non-zero current value (reference :586-632)."""
    differences: Dict[str, Any] = {}
    flags: List[str] = []
    for metric in REGRESSION_METRICS:
        cur, base = current.get(metric, 0.0), baseline.get(metric, 0.0)
        if base < 1e-6:
            if cur > 1e-6:
                flags.append(f"{metric}: {cur:.4f} (baseline was 0)")
            continue
        pct = 100.0 * (cur - base) / base
        differences[metric] = {"current": cur, "baseline": base, "change_percent": pct}
        if abs(pct) > threshold_percent:
            flags.append(f"{metric}: {pct:+.1f}% change (current={cur:.4f}, baseline={base:.4f})")
    return {"passed": not flags, "differences": differences, "flags": flags}


def compare_against_baseline(results: List[Dict[str, Any]], baseline_path: Path, threshold_percent: float = 10.0) -> bool:
    bar = "=" * 60
    print(f"\n{bar}\nRegression Testing: Comparing Against Baseline\n{bar}")
    baseline = load_baseline(baseline_path)
    if not baseline:
        print("No baseline found. Run with --update-baseline to create one.")
        return True
    base_patterns = baseline.get("patterns", {})
    flagged = []
    for r in results:
        name = r["pattern_name"]
        if name not in base_patterns:
            print(f"\n  {name}: Not in baseline (skipping)")
            continue
        for key, label, lead in (("single_scale", "Single-Scale", "\n"), ("pyramidal", "Pyramidal", "")):
            print(f"{lead}{name} ({label}):")
            cmp_ = compare_metrics(r[key]["metrics"], base_patterns[name][key]["metrics"], threshold_percent)
            if cmp_["passed"]:
                print("  Pass")
            else:
                print("  Regression detected:")
                for fl in cmp_["flags"]:
                    print(f"    - {fl}")
                flagged.append((name, label.lower(), cmp_["flags"]))
    print("\n" + bar)
    if not flagged:
        print("All patterns pass regression check")
    else:
        print(f"Regression detected in {len(flagged)} test(s)\n\nFlagged patterns:")
        for name, method, flags in flagged:
            print(f"  - {name} ({method}):")
            for fl in flags:
                print(f"      {fl}")
    print(bar)
    return not flagged


def update_baseline(results: List[Dict[str, Any]], baseline_path: Path) -> None:
    baseline_path = Path(baseline_path)
    baseline_path.parent.mkdir(parents=True, exist_ok=True)
    with open(baseline_path, "w") as f:
        json.dump(_results_document(results, datetime.now().isoformat()), f, indent=2)
    print(f"\nBaseline updated: {baseline_path}")


# ---- CLI -----------------------------------------------------------------------
def main() -> None:
    ap = argparse.ArgumentParser(description="Verify optical flow implementations against test suite")
    ap.add_argument("--config", type=str, default="python/verification_config.yaml")
    ap.add_argument("--pattern", type=str, nargs="+", help="Specific pattern(s) to test (default: all)")
    ap.add_argument("--pyramid-config", type=str, default="default")
    ap.add_argument("--no-visualizations", action="store_true")
    ap.add_argument("--compare-baseline", action="store_true")
    ap.add_argument("--update-baseline", action="store_true")
    ap.add_argument("--regression-threshold", type=float, default=10.0)
    ap.add_argument("--device-metrics", action="store_true",
                    help="reduce MAE/RMSE/EPE/AAE on the GPU (not a reference option; ~1e-6 relative to the host values)")
    args = ap.parse_args()

    config_path = Path(args.config)
    if not config_path.exists():
        print(f"Error: Config file not found: {config_path}")
        return
    config = load_config(config_path)
    bar = "=" * 60
    print(f"{bar}\nOptical Flow Verification Suite\n{bar}\nConfig: {config_path}\nPyramid config: {args.pyramid_config}")
    suite_dir = Path(config["test_suite_dir"])
    if not suite_dir.exists():
        print(f"\nError: Test suite not found: {suite_dir}\nRun generate_test_suite.py first to create test patterns.")
        return
    index = load_test_suite_index(suite_dir)
    print(f"Test suite: {suite_dir}\nPatterns available: {len(index['patterns'])}")
    names = list(index["patterns"].keys())
    if args.pattern:
        for p in args.pattern:
            if p not in index["patterns"]:
                print(f"Warning: Pattern '{p}' not found in suite\nAvailable patterns: {', '.join(sorted(names))}")
                return
        names = args.pattern
    print(f"Testing {len(names)} patterns\n")

    results = []
    for name in names:
        data = load_test_pattern(suite_dir / name)
        results.append(verify_pattern(name, data, config, pyramid_config_name=args.pyramid_config, verbose=True,
                                      device_metrics=args.device_metrics))
        if not args.no_visualizations and name in config["visualization"]["showcase_patterns"]:
            single = run_single_scale_lk(data["frame_prev"], data["frame_curr"],
                                         config["pyramids"]["default"]["window_size"])
            pyr = run_pyramidal_lk(data["frame_prev"], data["frame_curr"], config["pyramids"][args.pyramid_config])
            generate_visualizations(name, data, *single, *pyr, Path(config["output"]["visualizations_dir"]), config)

    print(f"\n{bar}\nGenerating Output Reports\n{bar}")
    table = generate_markdown_table(results)
    print("\n" + table)
    md_path = Path(config["output"]["results_markdown"])
    md_path.parent.mkdir(parents=True, exist_ok=True)
    md_path.write_text(table)
    print(f"\nMarkdown table saved: {md_path}")
    save_results_json(results, Path(config["output"]["results_json"]))

    baseline_path = Path("python/verification_baseline.json")
    if args.update_baseline:
        update_baseline(results, baseline_path)
    if args.compare_baseline:
        if not compare_against_baseline(results, baseline_path, threshold_percent=args.regression_threshold):
            print("\n   Regression detected! Review changes before committing.")
            sys.exit(1)
    print(f"\n{bar}\nVerification Complete!\n{bar}")


if __name__ == "__main__":
    main()

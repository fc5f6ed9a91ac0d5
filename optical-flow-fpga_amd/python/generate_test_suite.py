#!/usr/bin/env python3
"""Test-pattern generator: counterpart of the reference's python/generate_test_suite.py.

Produces the same 13 named frame pairs (same files: frame_0x.bin/.mem/.png,
metadata.json, suite_index.json) the reference's verifier consumes, WITHOUT
OpenCV: ``cv2.getRotationMatrix2D`` + ``cv2.warpAffine(INTER_LINEAR,
BORDER_CONSTANT, 128)`` (reference generate_test_suite.py:165-204) are restated
in NumPy, including warpAffine's fixed-point coordinate and weight arithmetic, so
the bytes are the ones the reference's baseline numbers were made from (checked
in tests/test_harness.py against the digests recorded in SURVEY.md Appendix B
and, through the metrics, against python/verification_baseline.json).

Host-side fixture generation only; nothing here is on the GPU hot path.
"""
from __future__ import annotations

import argparse
import json
import math
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np

SCRIPT_DIR = Path(__file__).resolve().parent
REPO_ROOT = SCRIPT_DIR.parents[1]
TEST_SUITE_DIR = SCRIPT_DIR / "test_suite"
# the reference keeps its texture at python/test_data/mountain_texture.jpg; this
# repo carries that data file as a test fixture
TEXTURE_CANDIDATES = [SCRIPT_DIR / "test_data" / "mountain_texture.jpg",
                      REPO_ROOT / "tests" / "golden" / "mountain_texture.jpg"]


@dataclass
class MotionParameters:
    """Ground-truth motion of one pattern (reference :41-54)."""

    name: str
    dx: float = 0.0
    dy: float = 0.0
    rotation: float = 0.0  # degrees, counter-clockwise
    scale: float = 1.0
    description: str = ""

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


def _mp(name, description, **kw) -> MotionParameters:
    return MotionParameters(name=name, description=description, **kw)


# the reference's suite, in its order (reference :57-137)
TEST_PATTERNS: Dict[str, MotionParameters] = {p.name: p for p in [
    _mp("translate_small", "Sub-pixel motion (tests fixed-point precision)", dx=0.5, dy=0.5),
    _mp("translate_medium", "Medium horizontal motion (standard test case)", dx=2.0),
    _mp("translate_large", "Large motion (challenges single-scale L-K)", dx=15.0),
    _mp("translate_vertical", "Vertical motion test", dy=10.0),
    _mp("translate_diagonal", "Diagonal motion (tests both components)", dx=10.0, dy=10.0),
    _mp("rotate_small", "Small rotation (2°) - violates brightness constancy", rotation=2.0),
    _mp("rotate_medium", "Medium rotation (5°) - tests algorithm limits", rotation=5.0),
    _mp("rotate_large", "Large rotation (15°) - expected failure for L-K", rotation=15.0),
    _mp("zoom_in", "Zoom in (10% expansion)", scale=1.1),
    _mp("zoom_out", "Zoom out (10% contraction)", scale=0.9),
    _mp("translate_rotate", "Combined translation + rotation", dx=5.0, dy=5.0, rotation=3.0),
    _mp("no_motion", "Stationary pattern (sanity check - expect zero flow)"),
    _mp("translate_extreme", "Extreme motion (far beyond window size)", dx=30.0, dy=20.0),
]}


def find_texture() -> Path:
    for c in TEXTURE_CANDIDATES:
        if c.exists():
            return c
    raise FileNotFoundError("Base texture mountain_texture.jpg not found in: "
                            + ", ".join(str(c) for c in TEXTURE_CANDIDATES))


def load_base_texture(width: int = 320, height: int = 240) -> np.ndarray:
    """Grayscale uint8 base frame: JPEG -> L -> bilinear resize (reference :140-162)."""
    from PIL import Image

    img = Image.open(find_texture()).convert("L").resize((width, height), Image.Resampling.BILINEAR)
    return np.array(img, dtype=np.uint8)


def rotation_matrix_2d(center: Tuple[float, float], angle_deg: float, scale: float) -> np.ndarray:
    """cv2.getRotationMatrix2D: 2x3 float64."""
    a = math.radians(angle_deg)
    alpha, beta = scale * math.cos(a), scale * math.sin(a)
    cx, cy = center
    return np.array([[alpha, beta, (1.0 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1.0 - alpha) * cy]], dtype=np.float64)


def warp_affine_linear_u8(src: np.ndarray, M: np.ndarray, border_value: int = 128) -> np.ndarray:
    """cv2.warpAffine(src, M, (W, H), INTER_LINEAR, BORDER_CONSTANT, border_value) for uint8.

    OpenCV inverts M, walks the destination grid with 10-bit fixed-point source
    coordinates, keeps 5 fractional bits, and blends with 15-bit integer weights.
    """
    H, W = src.shape
    m = M.astype(np.float64).copy()
    # invert the forward map exactly the way cv::warpAffine does
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0.0 else 0.0
    a11, a22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = a11, m[0, 1] * -D, m[1, 0] * -D, a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2

    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    ROUND_DELTA = AB_SCALE // (1 << INTER_BITS) // 2  # 16
    xs = np.arange(W, dtype=np.float64)
    ys = np.arange(H, dtype=np.float64)
    adelta = np.rint(m[0, 0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(m[1, 0] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((m[0, 1] * ys + m[0, 2]) * AB_SCALE).astype(np.int64) + ROUND_DELTA
    Y0 = np.rint((m[1, 1] * ys + m[1, 2]) * AB_SCALE).astype(np.int64) + ROUND_DELTA
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx, sy = X >> INTER_BITS, Y >> INTER_BITS
    fx, fy = X & 31, Y & 31

    # 15-bit bilinear weights; (fx, fy) = (0, 0) saturates to 32767 and the table
    # fix-up puts the missing 1 on the last tap
    w00 = (32 - fy) * (32 - fx) * 32
    w01 = (32 - fy) * fx * 32
    w10 = fy * (32 - fx) * 32
    w11 = fy * fx * 32
    origin = (fx == 0) & (fy == 0)
    w00 = np.where(origin, 32767, w00)
    w11 = np.where(origin, 1, w11)

    def tap(yy, xx):
        inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        vals = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return np.where(inside, vals, border_value)

    acc = (w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1))
    out = (acc + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def apply_motion_affine(frame: np.ndarray, params: MotionParameters) -> np.ndarray:
    """scale -> rotate about the centre -> translate (reference :165-204)."""
    height, width = frame.shape
    M = rotation_matrix_2d((width / 2.0, height / 2.0), params.rotation, params.scale)
    M[0, 2] += params.dx
    M[1, 2] += params.dy
    return warp_affine_linear_u8(frame, M, border_value=128)


# the reference's name for this step
apply_motion_opencv = apply_motion_affine


def _write_mem(path: Path, frame: np.ndarray) -> None:
    with open(path, "w") as f:
        f.write("".join(f"{int(v):02x}\n" for v in frame.ravel()))


def generate_test_pattern(params: MotionParameters, width: int = 320, height: int = 240,
                          output_dir: Optional[Path] = None, save_mem: bool = True, save_bin: bool = True,
                          save_png: bool = True, base: Optional[np.ndarray] = None
                          ) -> Tuple[np.ndarray, np.ndarray]:
    """One frame pair + metadata.json (reference :207-285)."""
    frame_0 = load_base_texture(width, height) if base is None else base
    frame_1 = apply_motion_affine(frame_0, params)
    if output_dir is not None:
        pdir = Path(output_dir) / params.name
        pdir.mkdir(parents=True, exist_ok=True)
        rigid = params.rotation == 0 and params.scale == 1.0
        metadata = {
            "pattern_name": params.name,
            "description": params.description,
            "resolution": {"width": width, "height": height},
            "motion_parameters": params.to_dict(),
            "expected_flow": {
                "u_mean": params.dx if rigid else "variable",
                "v_mean": params.dy if rigid else "variable",
                "note": "For rotation/zoom, flow varies spatially. Use test regions.",
            },
        }
        (pdir / "metadata.json").write_text(json.dumps(metadata, indent=2))
        if save_bin:
            frame_0.tofile(pdir / "frame_00.bin")
            frame_1.tofile(pdir / "frame_01.bin")
        if save_mem:
            _write_mem(pdir / "frame_00.mem", frame_0)
            _write_mem(pdir / "frame_01.mem", frame_1)
        if save_png:
            from PIL import Image

            Image.fromarray(frame_0).save(pdir / "frame_00.png")
            Image.fromarray(frame_1).save(pdir / "frame_01.png")
            Image.fromarray(np.hstack([frame_0, frame_1])).save(pdir / "comparison.png")
        print(f"  Generated: {params.name}")
        print(f"    Motion: dx={params.dx:.1f}, dy={params.dy:.1f}, rot={params.rotation:.1f}°, "
              f"scale={params.scale:.2f}")
    return frame_0, frame_1


def write_suite_index(output_dir: Path, width: int, height: int) -> None:
    index = {
        "suite_name": "Optical Flow Verification Suite",
        "resolution": {"width": width, "height": height},
        "num_patterns": len(TEST_PATTERNS),
        "patterns": {name: p.to_dict() for name, p in TEST_PATTERNS.items()},
    }
    (Path(output_dir) / "suite_index.json").write_text(json.dumps(index, indent=2))


def generate_full_suite(width: int = 320, height: int = 240, output_dir: Optional[Path] = None,
                        base: Optional[np.ndarray] = None, save_png: bool = True) -> None:
    """All 13 patterns + suite_index.json (reference :288-336)."""
    output_dir = Path(output_dir) if output_dir is not None else TEST_SUITE_DIR
    output_dir.mkdir(parents=True, exist_ok=True)
    bar = "=" * 60
    print(f"{bar}\nGenerating Optical Flow Test Suite\n{bar}")
    print(f"Resolution: {width}x{height}\nOutput directory: {output_dir}\nNumber of patterns: {len(TEST_PATTERNS)}\n")
    if base is None:
        base = load_base_texture(width, height)
    for params in TEST_PATTERNS.values():
        generate_test_pattern(params, width, height, output_dir, base=base, save_png=save_png)
    write_suite_index(output_dir, width, height)
    print(f"\n{bar}\nTest Suite Generation Complete\n{bar}")
    print(f"Generated {len(TEST_PATTERNS)} test patterns\nSuite index: {output_dir / 'suite_index.json'}\n")


def main() -> None:
    ap = argparse.ArgumentParser(description="Generate optical flow test patterns with known ground truth")
    ap.add_argument("--pattern", type=str, default="all", help='"all", "custom" or a pattern name')
    ap.add_argument("--list", action="store_true", help="List available test patterns")
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--output-dir", type=str, default=None)
    ap.add_argument("--dx", type=float, default=0.0)
    ap.add_argument("--dy", type=float, default=0.0)
    ap.add_argument("--rotation", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()

    if args.list:
        print("\nAvailable Test Patterns:\n" + "-" * 60)
        for name, p in TEST_PATTERNS.items():
            print(f"{name:25s} - {p.description}")
        print("")
        return
    out = Path(args.output_dir) if args.output_dir else TEST_SUITE_DIR
    if args.pattern == "all":
        generate_full_suite(args.width, args.height, out)
    elif args.pattern == "custom":
        p = MotionParameters(name="custom", dx=args.dx, dy=args.dy, rotation=args.rotation, scale=args.scale,
                             description=f"Custom: dx={args.dx}, dy={args.dy}, rot={args.rotation}°")
        print(f"Generating custom pattern: {p.description}")
        generate_test_pattern(p, args.width, args.height, out)
        print(f"Saved to: {out / 'custom'}")
    elif args.pattern in TEST_PATTERNS:
        p = TEST_PATTERNS[args.pattern]
        print(f"Generating pattern: {p.name}\n  {p.description}")
        generate_test_pattern(p, args.width, args.height, out)
        print(f"Saved to: {out / p.name}")
    else:
        print(f"ERROR: Unknown pattern '{args.pattern}'\nUse --list to see available patterns")
        raise SystemExit(1)


if __name__ == "__main__":
    main()

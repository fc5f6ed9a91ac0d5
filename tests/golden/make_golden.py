#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the outputs are
committed, the reference never travels.  What is written is data only:

  patterns_320x240.npz       the 13 test-suite frame pairs (uint8), regenerated
                             cv2-free by optical-flow-fpga_amd/python/generate_test_suite.py
  reference_13patterns.json  per pattern, from the reference's own functions:
                             sha256 of the single-scale and pyramidal (u, v) bytes
                             (after +0.0, which maps -0.0 to +0.0), the five
                             metrics of flow_metrics.compute_all_metrics on the
                             verifier's mask, iterations run per level, residual log
  dense_translate_medium.npz full reference flow fields of one pattern
  stage_vectors.npz          per-stage vectors on small inputs (gradients, window
                             solve for 3/5/7, pyramid, warp, upsample, pyramidal)
  rtl_frames.npz             the two frame pairs the reference commits as .mem
                             (tb/test_frames, python/tb/test_frames) + reference flows' region means

Usage:  python tests/golden/make_golden.py          (about 3 minutes)
"""
from __future__ import annotations

import contextlib
import hashlib
import io
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = Path("/root/reference")

sys.path.insert(0, str(REF / "python"))
import flow_metrics as R_metrics  # noqa: E402  (reference)
import lucas_kanade_core as R_core  # noqa: E402  (reference)
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)

# the reference dumps PNGs from inside the hot function (:226); not wanted here
R_pyr.visualize_pyramid_level = lambda *a, **k: None

# the verifier's mask logic, imported from the reference too (pure function)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_verifier", REF / "python" / "optical_flow_verifier.py")
R_ver = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(R_ver)

# this repo's cv2-free generator (loaded by path: its module name also exists in the reference)
_gspec = importlib.util.spec_from_file_location(
    "oflk_generate_test_suite", ROOT / "optical-flow-fpga_amd" / "python" / "generate_test_suite.py")
G = importlib.util.module_from_spec(_gspec)
sys.modules["oflk_generate_test_suite"] = G
_gspec.loader.exec_module(G)


def digest(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)  # -0.0 -> +0.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def ref_pyramidal_with_log(prev, curr, levels=3, win=5, iters=3):
    """Run the reference's pyramidal LK, capturing the residual means it prints."""
    rec = []
    orig = R_pyr.lucas_kanade_single_scale

    def spy(a, b, w):
        du, dv = orig(a, b, w)
        rec.append((a.shape, np.mean(np.abs(du)), np.mean(np.abs(dv))))
        return du, dv

    R_pyr.lucas_kanade_single_scale = spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            u, v = R_pyr.lucas_kanade_pyramidal(prev, curr, levels, win, iters)
    finally:
        R_pyr.lucas_kanade_single_scale = orig
    shapes = []
    for s, _, _ in rec:
        if not shapes or shapes[-1] != s:
            shapes.append(s)
    runs = [sum(1 for s, _, _ in rec if s == sh) for sh in shapes]
    log = [[[float(mu), float(mv)] for s, mu, mv in rec if s == sh] for sh in shapes]
    return u, v, runs, log


def load_mem(path: Path, h=240, w=320) -> np.ndarray:
    return np.array([int(t, 16) for t in path.read_text().split()], np.uint8).reshape(h, w)


def main() -> None:
    t0 = time.time()
    cfg_crop = 80
    base = G.load_base_texture(320, 240)
    frames1 = {}
    out = {"generator": "tests/golden/make_golden.py", "reference": "rothej/optical-flow-fpga @ 2026-02-20",
           "numpy": np.__version__, "scipy": __import__("scipy").__version__,
           "frame0_sha256": hashlib.sha256(base.tobytes()).hexdigest(), "patterns": {}}
    dense = {}
    for name, params in G.TEST_PATTERNS.items():
        f0, f1 = G.generate_test_pattern(params, base=base)
        frames1[name] = f1
        prev, curr = f0.astype(np.float32), f1.astype(np.float32)
        us, vs = R_core.lucas_kanade_single_scale(prev, curr, 5)
        up, vp, runs, log = ref_pyramidal_with_log(prev, curr, 3, 5, 3)
        mask = R_ver.get_test_region_mask(prev.shape, name, cfg_crop)
        out["patterns"][name] = {
            "frame1_sha256": hashlib.sha256(f1.tobytes()).hexdigest(),
            "motion": params.to_dict(),
            "single_scale": {"u_sha256": digest(us), "v_sha256": digest(vs),
                             "metrics": R_metrics.compute_all_metrics(us, vs, params.dx, params.dy, mask)},
            "pyramidal": {"u_sha256": digest(up), "v_sha256": digest(vp), "iters_run": runs, "residual_log": log,
                          "metrics": R_metrics.compute_all_metrics(up, vp, params.dx, params.dy, mask)},
            "num_test_pixels": int(mask.sum()),
        }
        if name == "translate_medium":
            dense = {"single_u": us, "single_v": vs, "pyr_u": up, "pyr_v": vp}
        print(f"{name:20s} single+pyramidal done, iters {runs}  [{time.time() - t0:.0f}s]", flush=True)

    np.savez_compressed(HERE / "patterns_320x240.npz", frame_0=base, **{f"frame_1__{k}": v for k, v in frames1.items()})
    (HERE / "reference_13patterns.json").write_text(json.dumps(out, indent=1))
    np.savez_compressed(HERE / "dense_translate_medium.npz", **dense)

    # ---- per-stage vectors on small inputs ---------------------------------------
    rng = np.random.default_rng(20260220)
    sv = {}
    f1 = frames1["translate_medium"]
    tile_p = base[96:144, 128:192].astype(np.float32)          # 48 x 64, integer-valued
    tile_c = f1[96:144, 128:192].astype(np.float32)
    nz_p = rng.normal(110, 40, (45, 61)).astype(np.float32)    # non-integer, odd sizes
    nz_c = (nz_p + rng.normal(0, 5, (45, 61))).astype(np.float32)
    for tag, (p, c) in {"tile": (tile_p, tile_c), "noise": (nz_p, nz_c)}.items():
        sv[f"{tag}__prev"], sv[f"{tag}__curr"] = p, c
        Ix, Iy, It = R_core.compute_gradients(p, c)
        sv[f"{tag}__Ix"], sv[f"{tag}__Iy"], sv[f"{tag}__It"] = Ix, Iy, It
        for win in (3, 4, 5, 7):
            u, v = R_core.lucas_kanade_single_scale(p, c, win)
            sv[f"{tag}__single_u_w{win}"], sv[f"{tag}__single_v_w{win}"] = u, v
        pyr = R_pyr.build_gaussian_pyramid(p, 3)
        for l, a in enumerate(pyr):
            sv[f"{tag}__pyr{l}"] = a
        fu = rng.normal(0, 3, p.shape).astype(np.float32)
        fv = rng.normal(0, 3, p.shape).astype(np.float32)
        fu[3, :] = np.float32(p.shape[1] - 1) - np.arange(p.shape[1], dtype=np.float32)  # lands on x = W-1 exactly
        fv[:, 4] = -np.arange(p.shape[0], dtype=np.float32)                               # lands on y = 0 exactly
        sv[f"{tag}__flow_u"], sv[f"{tag}__flow_v"] = fu, fv
        sv[f"{tag}__warped"] = R_pyr.warp_image(c, fu, fv)
        uu, vv = R_pyr.upsample_flow(fu, fv, (2 * p.shape[0] + 1, 2 * p.shape[1]))
        sv[f"{tag}__up_u"], sv[f"{tag}__up_v"] = uu, vv
        u2, v2, runs, log = ref_pyramidal_with_log(p, c, 2, 5, 3)
        sv[f"{tag}__pyrlk_u"], sv[f"{tag}__pyrlk_v"] = u2, v2
        sv[f"{tag}__pyrlk_runs"] = np.array(runs, np.int32)
        sv[f"{tag}__pyrlk_log"] = np.array([r + [[0.0, 0.0]] * (3 - len(r)) for r in log], np.float32)
    # np.mean(np.abs(x)) known answers (fp32 pairwise, 8192-element pieces)
    for n in (25, 4800, 8192, 8200, 76800):
        x = rng.normal(0, 1, n).astype(np.float32)
        sv[f"meanabs__x{n}"] = x
        sv[f"meanabs__y{n}"] = np.array([np.mean(np.abs(x))], np.float32)
    np.savez_compressed(HERE / "stage_vectors.npz", **sv)

    # ---- the frame pairs the reference commits for its RTL testbench ---------------
    rtl = {}
    for tag, d in {"natural": REF / "tb" / "test_frames", "sinusoid": REF / "python" / "tb" / "test_frames"}.items():
        a, b = load_mem(d / "frame_00.mem"), load_mem(d / "frame_01.mem")
        rtl[f"{tag}__frame_00"], rtl[f"{tag}__frame_01"] = a, b
        p, c = a.astype(np.float32), b.astype(np.float32)
        us, vs = R_core.lucas_kanade_single_scale(p, c, 5)
        up, vp, runs, log = ref_pyramidal_with_log(p, c, 3, 5, 3)
        reg = np.s_[105:135, 55:85]  # lucas_kanade_reference.py:171
        rtl[f"{tag}__answers"] = np.array([np.mean(us[reg]), np.mean(vs[reg]), np.mean(up[reg]), np.mean(vp[reg]),
                                           float(np.sum(us != 0))], np.float64)
        rtl[f"{tag}__sha"] = np.array([digest(us), digest(vs), digest(up), digest(vp)])
        rtl[f"{tag}__runs"] = np.array(runs, np.int32)
    np.savez_compressed(HERE / "rtl_frames.npz", **rtl)
    print(f"done in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()

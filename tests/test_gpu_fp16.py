"""BASELINE.json config 5 (7680x4320 pair, 7x7 window, fp16 gradients / accumulators): the opt-in
reduced-precision single-scale mode against the exact fp32 path.

The reference has no such mode (its arithmetic is fp32, lucas_kanade_core.py:110-133), so the bar is
not equality: SURVEY.md section 7 -- "parity target there is EPE vs fp32 reference REPORTED, not 1e-4".
The endpoint error (EPE) of the fp16 flow against the exact flow is measured per pattern and held to
the tolerances below, which are properties of half precision (11-bit significands in the gradients and
in 49-tap sums), not tuning knobs:

  * well-conditioned pixels (|det| of the exact normal matrix among the upper half of the frame's
    values): mean EPE <= 0.01 px   (measured: <= 0.0042 px over the 13 patterns, 5x5 and 7x7)
  * all pixels: MEDIAN EPE <= 0.01 px (measured: <= 0.0023 px; the mean over all pixels -- measured
    <= 0.0072 px -- is reported but not bounded: where
    det ~ 0 the exact flow itself reaches thousands of pixels -- 7078 px on translate_extreme -- and any
    rounding moves it by pixels)
  * the mode keeps the reference's border and det-threshold semantics: borders exactly 0.

Numbers of one run are written to gpurun_out/fp16_epe.json (copied to profiles/ by the refresh script).
"""
import json
import os
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
PATTERNS = ["translate_small", "translate_medium", "translate_large", "translate_vertical", "translate_diagonal",
            "rotate_small", "rotate_medium", "rotate_large", "zoom_in", "zoom_out", "translate_rotate", "no_motion",
            "translate_extreme"]
TOL_WELL_CONDITIONED_MEAN = 0.01   # px
TOL_MEDIAN = 0.01                  # px

_report = {}


def _dets(p, c, win):
    """det of the exact (float64) normal matrix per pixel, to tell well- from ill-conditioned windows"""
    from scipy.ndimage import uniform_filter
    from scipy.signal import convolve2d

    avg = (p.astype(np.float64) + c) / 2
    sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]) / 8.0
    ix = convolve2d(avg, sx, mode="same", boundary="symm")
    iy = convolve2d(avg, sx.T, mode="same", boundary="symm")
    n = win * win
    sxx, syy, sxy = (uniform_filter(q, win, mode="constant") * n for q in (ix * ix, iy * iy, ix * iy))
    return sxx * syy - sxy * sxy


def _epe_stats(p, c, win):
    import lucas_kanade_core as K

    u, v = K.lucas_kanade_single_scale(p, c, win)
    hu, hv = K.lucas_kanade_single_scale_fp16(p, c, win, 255.0)
    assert np.isfinite(hu).all() and np.isfinite(hv).all()
    hw = win // 2
    # borders: exactly zero, like the reference (lucas_kanade_core.py:101-108)
    for a in (hu, hv):
        assert not a[:hw].any() and not a[-hw:].any() and not a[:, :hw].any() and not a[:, -hw:].any()
    inner = (slice(hw, -hw), slice(hw, -hw))
    epe = np.sqrt((hu.astype(np.float64) - u) ** 2 + (hv.astype(np.float64) - v) ** 2)[inner]
    det = _dets(p, c, win)[inner]
    good = det >= np.median(det)
    return {"mean_epe_all": float(epe.mean()), "median_epe_all": float(np.median(epe)),
            "mean_epe_well_conditioned": float(epe[good].mean()), "p99_epe_well_conditioned": float(np.percentile(epe[good], 99)),
            "max_abs_exact_flow": float(max(np.abs(u).max(), np.abs(v).max()))}


@pytest.fixture(scope="module")
def suite(golden_dir):
    return np.load(golden_dir / "patterns_320x240.npz")


@pytest.mark.parametrize("name", PATTERNS)
@pytest.mark.parametrize("win", [7, 5])
def test_fp16_epe_on_the_13_patterns(suite, name, win):
    p, c = suite["frame_0"].astype(np.float32), suite[f"frame_1__{name}"].astype(np.float32)
    st = _epe_stats(p, c, win)
    _report[f"{name} {win}x{win}"] = st
    assert st["median_epe_all"] <= TOL_MEDIAN, st
    assert st["mean_epe_well_conditioned"] <= TOL_WELL_CONDITIONED_MEAN, st


def test_fp16_at_8k_config5():
    """the config as stated: 7680x4320 pair, 7x7 window; crops against the exact path (a full 8K float64
    conditioning map is not needed: the synthetic frames are textured everywhere)"""
    import torch

    import _oflk
    from oflk_synth import synth_pair

    H, W, win = 4320, 7680, 7
    p, c = synth_pair(H, W, 0)
    dev = torch.device("cuda", 0)
    tp, tc = torch.from_numpy(p).to(dev), torch.from_numpy(c).to(dev)
    u, v, hu, hv = (torch.empty_like(tp) for _ in range(4))
    plan = _oflk.Plan(0, 1, H, W, 1, win, 0)
    st = torch.cuda.current_stream().cuda_stream
    plan.single_scale(tp.data_ptr(), tc.data_ptr(), u.data_ptr(), v.data_ptr(), st)
    plan.single_scale_fp16(tp.data_ptr(), tc.data_ptr(), hu.data_ptr(), hv.data_ptr(), 255.0, st)
    torch.cuda.synchronize()
    assert torch.isfinite(hu).all() and torch.isfinite(hv).all()
    epe = torch.sqrt((hu.double() - u.double()) ** 2 + (hv.double() - v.double()) ** 2)[3:-3, 3:-3]
    stats = {"mean_epe_all": float(epe.mean()), "median_epe_all": float(epe.median()),
             "p99_epe_all": float(torch.quantile(epe.flatten()[::97].float(), 0.99))}
    # timing, inputs resident: the figure profiles/<tag>_configs.json carries
    for fn, key in ((plan.single_scale, "exact_fp32_us"), (plan.single_scale_fp16, "fp16_us")):
        args = (tp.data_ptr(), tc.data_ptr(), hu.data_ptr(), hv.data_ptr()) + ((255.0, st) if key == "fp16_us" else (st,))
        for _ in range(3):
            fn(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        stats[key] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    stats["fp16_GBs_algorithmic"] = round(16.0 * H * W / stats["fp16_us"] / 1e3, 1)
    _report["8K 7x7 synthetic"] = stats
    plan.close()
    assert stats["median_epe_all"] <= TOL_MEDIAN, stats


def test_fp16_other_windows_and_ragged_shapes():
    import lucas_kanade_core as K

    rng = np.random.default_rng(4)
    for (H, W) in ((37, 53), (64, 64), (5, 300), (131, 70)):
        a = rng.integers(0, 256, (H, W)).astype(np.float32)
        b = np.roll(a, (1, 1), (0, 1))
        for win in (3, 5, 7, 9, 11):
            u, v = K.lucas_kanade_single_scale(a, b, win)
            hu, hv = K.lucas_kanade_single_scale_fp16(a, b, win)
            assert hu.shape == (H, W) and np.isfinite(hu).all() and np.isfinite(hv).all()
            if min(H, W) > win:
                epe = np.sqrt((hu - u) ** 2 + (hv - v) ** 2)
                assert np.median(epe) <= 0.1, (H, W, win, float(np.median(epe)))


def test_fp16_one_and_two_columns_per_lane_agree(suite):
    """the streaming kernel exists with two columns per lane (k_lk16d, the one that runs) and with one (k_lk16s,
    OFLK_LK16_COLS=1).  Same arithmetic contract; the packed sums are added in a different
    order across columns, so the flows agree to fp16 rounding of the window sums, not bit for bit.  The forcing switch
    is read once per process: the one-column form runs in a child process."""
    import subprocess
    import sys

    import lucas_kanade_core as K

    p = suite["frame_0"].astype(np.float32)
    c = suite["frame_1__rotate_medium"].astype(np.float32)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import lucas_kanade_core as K; z = np.load(%r); "
            "u, v = K.lucas_kanade_single_scale_fp16(z['frame_0'].astype(np.float32), z['frame_1__rotate_medium'].astype(np.float32), 7, 255.0); "
            "np.save(sys.argv[1], np.stack([u, v]))")
    out = ROOT / "gpurun_out" / "fp16_cols1.npy"
    out.parent.mkdir(exist_ok=True)
    env = dict(os.environ, OFLK_LK16_COLS="1")
    subprocess.run([sys.executable, "-c", code % (str(ROOT / "optical-flow-fpga_amd" / "python"), str(ROOT / "tests" / "golden" / "patterns_320x240.npz")), str(out)],
                   check=True, env=env, timeout=300)
    u1, v1 = np.load(out)
    u2, v2 = K.lucas_kanade_single_scale_fp16(p, c, 7, 255.0)
    ue, ve = K.lucas_kanade_single_scale(p, c, 7)
    for hu, hv in ((u1, v1), (u2, v2)):
        assert np.median(np.sqrt((hu.astype(np.float64) - ue) ** 2 + (hv.astype(np.float64) - ve) ** 2)) <= TOL_MEDIAN
    assert np.median(np.sqrt((u1.astype(np.float64) - u2) ** 2 + (v1.astype(np.float64) - v2) ** 2)) <= TOL_MEDIAN
    out.unlink()


def test_fp16_tiled_form_agrees_with_the_streaming_form(suite, monkeypatch):
    """the library holds two kernels for this mode: the streaming one (default: one wave per 64-column strip,
    everything in registers) and the LDS-tiled one (OFLK_LK16_TILED=1, kept for its tile sizing);
    both meet the same tolerances and agree with each other to within fp16 rounding of the window sums"""
    import lucas_kanade_core as K

    p = suite["frame_0"].astype(np.float32)
    for name in ("translate_medium", "rotate_medium", "zoom_in"):
        c = suite[f"frame_1__{name}"].astype(np.float32)
        for win in (5, 7):
            monkeypatch.delenv("OFLK_LK16_TILED", raising=False)
            su, sv = K.lucas_kanade_single_scale_fp16(p, c, win, 255.0)
            monkeypatch.setenv("OFLK_LK16_TILED", "1")
            tu, tv = K.lucas_kanade_single_scale_fp16(p, c, win, 255.0)
            monkeypatch.delenv("OFLK_LK16_TILED")
            u, v = K.lucas_kanade_single_scale(p, c, win)
            for hu, hv in ((su, sv), (tu, tv)):
                epe = np.sqrt((hu.astype(np.float64) - u) ** 2 + (hv.astype(np.float64) - v) ** 2)
                assert np.median(epe) <= TOL_MEDIAN
            d = np.sqrt((su.astype(np.float64) - tu) ** 2 + (sv.astype(np.float64) - tv) ** 2)
            assert np.median(d) <= TOL_MEDIAN, (name, win, float(np.median(d)))


def test_fp16_strip_and_segment_seams():
    """the streaming kernel cuts the frame into 64 - 2R column strips and Hs-row segments: widths and heights
    around those seams, several pairs per call, and a forced small segment height give the same flow as one
    segment does (the arithmetic per pixel does not depend on the cut)"""
    import os

    import torch

    import _oflk

    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)
    # two columns per lane: strips of 124 (3x3), 120 (5x5, 7x7), 116 (9x9, 11x11) columns
    for (B, H, W, win) in ((2, 90, 56, 7), (1, 77, 113, 7), (3, 41, 58 * 3 + 1, 5), (1, 200, 129, 3), (2, 60, 241, 7), (1, 50, 121, 7),
                           (1, 64, 250, 11), (2, 33, 117, 9), (1, 45, 375, 3)):
        a = rng.integers(0, 256, (B, H, W)).astype(np.float32)
        b = np.roll(a, (1, -1), (1, 2))
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        outs = []
        for hs in (None, "8", "13"):
            if hs is None:
                os.environ.pop("OFLK_LK16_HS", None)
            else:
                os.environ["OFLK_LK16_HS"] = hs
            u, v = torch.full_like(ta, 7.0), torch.full_like(ta, 7.0)
            plan = _oflk.Plan(0, B, H, W, 1, win, 0)
            plan.single_scale_fp16(ta.data_ptr(), tb.data_ptr(), u.data_ptr(), v.data_ptr(), 255.0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            plan.close()
            outs.append((u.cpu().numpy(), v.cpu().numpy()))
        os.environ.pop("OFLK_LK16_HS", None)
        for u, v in outs[1:]:
            assert np.array_equal(u, outs[0][0]) and np.array_equal(v, outs[0][1]), (B, H, W, win)
        hw = win // 2
        assert not outs[0][0][:, :hw].any() and not outs[0][0][:, -hw:].any() and not outs[0][0][:, :, :hw].any() and not outs[0][0][:, :, -hw:].any()


def test_zz_write_report():
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "fp16_epe.json").write_text(json.dumps({"tolerances_px": {"median_all": TOL_MEDIAN,
                                                                      "mean_well_conditioned": TOL_WELL_CONDITIONED_MEAN},
                                                   "reference_for_epe": "exact fp32 path (equal to the Python reference value for value)",
                                                   "rows": _report}, indent=1))
    assert _report

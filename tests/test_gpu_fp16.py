"""BASELINE.json config 5 (7680x4320 pair, 7x7 window, fp16 gradients / accumulators): the opt-in
reduced-precision single-scale mode against the exact result -- on the 13 patterns the exact flow is the CPU ORACLE's
(pinned to the reference by tests/test_oracle_golden.py), so a regression shared by both HIP paths cannot hide.

The reference has no such mode (its arithmetic is fp32, lucas_kanade_core.py:110-133), so the bar is
not equality: SURVEY.md section 7 -- "parity target there is EPE vs fp32 reference REPORTED, not 1e-4".
The endpoint error (EPE) of the fp16 flow against the exact flow is measured per pattern and held to
the tolerances below, which are properties of half precision (11-bit significands in the gradients and
in 49-tap sums), not tuning knobs:

  * well-conditioned pixels (|det| of the exact normal matrix among the upper half of the frame's
    values): mean EPE <= 0.01 px   (measured: <= 0.0042 px over the 13 patterns, 5x5 and 7x7)
  * all pixels: MEDIAN EPE <= 0.01 px (measured: <= 0.0023 px) and MEAN EPE <= 0.02 px (measured: <= 0.0072 px; where
    det ~ 0 the exact flow itself reaches thousands of pixels -- 7078 px on translate_extreme -- and any rounding moves
    it by pixels, so the bound on the mean is a regression gate at ~3x the measured value, not a property of fp16)
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
TOL_MEAN_ALL = 0.02                # px: regression gate, ~3x what is measured

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


def _epe_stats(p, c, win, exact=None):
    import lucas_kanade_core as K

    u, v = exact if exact is not None else K.lucas_kanade_single_scale(p, c, win)
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
def test_fp16_epe_on_the_13_patterns(suite, oracle, name, win):
    p, c = suite["frame_0"].astype(np.float32), suite[f"frame_1__{name}"].astype(np.float32)
    st = _epe_stats(p, c, win, exact=oracle.lucas_kanade_single_scale(p, c, win))   # the oracle's exact flow
    _report[f"{name} {win}x{win}"] = st
    assert st["median_epe_all"] <= TOL_MEDIAN, st
    assert st["mean_epe_well_conditioned"] <= TOL_WELL_CONDITIONED_MEAN, st
    assert st["mean_epe_all"] <= TOL_MEAN_ALL, st


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


def test_fp16_strip_and_segment_seams():
    """the streaming kernel cuts the frame into strips of 128 - 4 ceil(R/2) columns and segments of Hs rows (Hs follows
    from the frame height and the batch size): widths around the strip seams, several pairs per call, and the same
    frames with 17 / 40 more rows appended (the segments are then cut at other rows) give the same flow wherever the
    window does not see the difference -- the arithmetic per pixel does not depend on the cut"""
    import torch

    import _oflk

    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)
    # two columns per lane: strips of 124 (3x3), 120 (5x5, 7x7), 116 (9x9, 11x11) columns
    for (B, H, W, win) in ((2, 90, 56, 7), (1, 77, 113, 7), (3, 41, 58 * 3 + 1, 5), (1, 200, 129, 3), (2, 60, 241, 7), (1, 50, 121, 7),
                           (1, 64, 250, 11), (2, 33, 117, 9), (1, 45, 375, 3), (1, 700, 130, 7)):
        hw = win // 2
        a = rng.integers(0, 256, (B, H + 40, W)).astype(np.float32)
        b = np.roll(a, (1, -1), (1, 2))
        outs = []
        for extra in (0, 17, 40):
            Hx = H + extra
            ta = torch.from_numpy(np.ascontiguousarray(a[:, :Hx])).to(dev)
            tb = torch.from_numpy(np.ascontiguousarray(b[:, :Hx])).to(dev)
            u, v = torch.full_like(ta, 7.0), torch.full_like(ta, 7.0)
            plan = _oflk.Plan(0, B, Hx, W, 1, win, 0)
            plan.single_scale_fp16(ta.data_ptr(), tb.data_ptr(), u.data_ptr(), v.data_ptr(), 255.0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            plan.close()
            outs.append((u.cpu().numpy(), v.cpu().numpy()))
        keep = H - hw - 1   # rows whose window and Sobel ring lie inside the shortest frame
        for u, v in outs[1:]:
            assert np.array_equal(u[:, :keep], outs[0][0][:, :keep]) and np.array_equal(v[:, :keep], outs[0][1][:, :keep]), (B, H, W, win)
        u0 = outs[0][0]
        assert not u0[:, :hw].any() and not u0[:, -hw:].any() and not u0[:, :, :hw].any() and not u0[:, :, -hw:].any()


def test_zz_write_report():
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "fp16_epe.json").write_text(json.dumps({"tolerances_px": {"median_all": TOL_MEDIAN,
                                                                      "mean_well_conditioned": TOL_WELL_CONDITIONED_MEAN},
                                                   "reference_for_epe": "exact fp32 path (equal to the Python reference value for value)",
                                                   "rows": _report}, indent=1))
    assert _report

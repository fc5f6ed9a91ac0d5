"""Flow accuracy metrics against a constant ground-truth vector: counterpart of the
reference's python/flow_metrics.py (same function names, arguments and return
values, same float32 array arithmetic so the numbers agree with
verification_baseline.json).  Host-side harness code, not on the GPU hot path.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import numpy.typing as npt

Mask = Optional[npt.NDArray[np.bool_]]


def _errors(u_pred, v_pred, u_true: float, v_true: float, mask: Mask):
    """Per-pixel error components over the masked pixels (float32 arrays)."""
    if mask is None:
        mask = np.ones_like(u_pred, dtype=bool)
    return u_pred[mask] - u_true, v_pred[mask] - v_true


def mean_absolute_error(u_pred, v_pred, u_true: float, v_true: float, mask: Mask = None) -> Tuple[float, float]:
    """(MAE_u, MAE_v) in pixels (reference flow_metrics.py:14-40)."""
    eu, ev = _errors(u_pred, v_pred, u_true, v_true, mask)
    return float(np.mean(np.abs(eu))), float(np.mean(np.abs(ev)))


def root_mean_square_error(u_pred, v_pred, u_true: float, v_true: float, mask: Mask = None) -> float:
    """RMSE of the error vector length (reference :43-70)."""
    eu, ev = _errors(u_pred, v_pred, u_true, v_true, mask)
    return float(np.sqrt(np.mean(eu**2 + ev**2)))


def endpoint_error(u_pred, v_pred, u_true: float, v_true: float, mask: Mask = None) -> float:
    """Average endpoint error (reference :73-103)."""
    eu, ev = _errors(u_pred, v_pred, u_true, v_true, mask)
    return float(np.mean(np.sqrt(eu**2 + ev**2)))


def angular_error(u_pred, v_pred, u_true: float, v_true: float, mask: Mask = None) -> float:
    """Average angle, in degrees, between (u, v, 1) vectors (reference :106-163)."""
    if mask is None:
        mask = np.ones_like(u_pred, dtype=bool)
    up, vp = u_pred[mask], v_pred[mask]
    ut, vt = np.full_like(up, u_true), np.full_like(vp, v_true)
    one_p, one_t = np.ones_like(up), np.ones_like(ut)
    # nothing moves and nothing was predicted: define the angle as zero
    if np.sqrt(u_true**2 + v_true**2) < 1e-6 and np.all(np.sqrt(up**2 + vp**2) < 1e-6):
        return 0.0
    len_p = np.sqrt(up**2 + vp**2 + one_p**2)
    len_t = np.sqrt(ut**2 + vt**2 + one_t**2)
    cosang = (up * ut + vp * vt + one_p * one_t) / (len_p * len_t)
    cosang = np.clip(cosang, -1.0, 1.0)
    return float(np.mean(np.rad2deg(np.arccos(cosang))))


def compute_all_metrics(u_pred, v_pred, u_true: float, v_true: float, mask: Mask = None) -> Dict[str, float]:
    """mae_u, mae_v, rmse, epe, aae (reference :166-201)."""
    mae_u, mae_v = mean_absolute_error(u_pred, v_pred, u_true, v_true, mask)
    return {
        "mae_u": mae_u,
        "mae_v": mae_v,
        "rmse": root_mean_square_error(u_pred, v_pred, u_true, v_true, mask),
        "epe": endpoint_error(u_pred, v_pred, u_true, v_true, mask),
        "aae": angular_error(u_pred, v_pred, u_true, v_true, mask),
    }


METRIC_NAMES = ("mae_u", "mae_v", "rmse", "epe", "aae")


def mask_rectangle(mask: npt.NDArray[np.bool_]) -> Tuple[int, int, int, int]:
    """(y0, y1, x0, x1) of a mask that is one filled rectangle (what the verifier's
    get_test_region_mask builds); ValueError otherwise."""
    ys, xs = np.flatnonzero(mask.any(axis=1)), np.flatnonzero(mask.any(axis=0))
    if ys.size == 0:
        return 0, 0, 0, 0
    y0, y1, x0, x1 = int(ys[0]), int(ys[-1]) + 1, int(xs[0]), int(xs[-1]) + 1
    if not mask[y0:y1, x0:x1].all():
        raise ValueError("mask is not a filled rectangle")
    return y0, y1, x0, x1


def compute_all_metrics_gpu(u_pred, v_pred, u_true: float, v_true: float, mask: Mask = None) -> Dict[str, float]:
    """compute_all_metrics with the reductions on the MI355X (oflk_flow_metrics); rectangular
    masks only.  Agrees with compute_all_metrics to ~1e-6 relative (fp64 sums vs fp32 pairwise)."""
    import ctypes

    import _oflk

    u, v = _oflk.as_f32(u_pred), _oflk.as_f32(v_pred)
    H, W = _oflk.same_shape(u, v)
    y0, y1, x0, x1 = (0, H, 0, W) if mask is None else mask_rectangle(np.asarray(mask, bool))
    ut, vt = np.array([u_true], np.float32), np.array([v_true], np.float32)
    out = np.zeros((1, 5), np.float64)
    _oflk.check(_oflk.lib().oflk_flow_metrics(_oflk.ptr(u), _oflk.ptr(v), 1, H, W, _oflk.ptr(ut), _oflk.ptr(vt),
                                              y0, y1, x0, x1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
    return dict(zip(METRIC_NAMES, (float(x) for x in out[0])))

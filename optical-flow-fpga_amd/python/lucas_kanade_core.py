"""Drop-in for the reference's ``python/lucas_kanade_core.py``: same module name,
same function names, signatures and defaults, but every function runs
hand-written HIP kernels on an MI355X through liboflk's C ABI.

Put this directory on ``sys.path`` / ``PYTHONPATH`` in place of the reference's
``python/`` directory and ``from lucas_kanade_core import ...`` keeps working
(reference import sites: optical_flow_verifier.py:19, lucas_kanade_reference.py:14,
lucas_kanade_pyramidal.py:15).

Outputs equal the reference's value for value (see DESIGN.md "Exactness").
No CPU fallback exists: without the built library or a GPU the calls raise.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import numpy.typing as npt

import _oflk


def compute_gradients(
    frame_prev: npt.NDArray[np.float32], frame_curr: npt.NDArray[np.float32]
) -> Tuple[npt.NDArray[np.float32], npt.NDArray[np.float32], npt.NDArray[np.float32]]:
    """Sobel/8 spatial gradients of the frame average and It = prev - curr.

    Replaces reference lucas_kanade_core.py:15-45 (oflk_compute_gradients).
    """
    p, c = _oflk.as_f32(frame_prev), _oflk.as_f32(frame_curr)
    H, W = _oflk.same_shape(p, c)
    Ix = np.empty((H, W), np.float32)
    Iy = np.empty((H, W), np.float32)
    It = np.empty((H, W), np.float32)
    _oflk.check(_oflk.lib().oflk_compute_gradients(_oflk.ptr(p), _oflk.ptr(c), H, W, _oflk.ptr(Ix),
                                                   _oflk.ptr(Iy), _oflk.ptr(It)))
    return Ix, Iy, It


def lucas_kanade_single_scale(
    frame_prev: npt.NDArray[np.float32],
    frame_curr: npt.NDArray[np.float32],
    window_size: int = 5,
) -> Tuple[npt.NDArray[np.float32], npt.NDArray[np.float32]]:
    """Dense single-scale Lucas-Kanade flow (u, v); one fused GPU kernel.

    Replaces reference lucas_kanade_core.py:48-70 (oflk_single_scale).
    """
    if _oflk.both_u8(frame_prev, frame_curr):
        # raw 8-bit frames: converted on the device (same values as .astype(np.float32) first)
        p, c = np.ascontiguousarray(frame_prev), np.ascontiguousarray(frame_curr)
        H, W = _oflk.same_shape(p, c)
        u = np.empty((H, W), np.float32)
        v = np.empty((H, W), np.float32)
        _oflk.check(_oflk.lib().oflk_single_scale_u8(p.ctypes.data, c.ctypes.data, 1, H, W, int(window_size),
                                                     _oflk.ptr(u), _oflk.ptr(v)))
        return u, v
    p, c = _oflk.as_f32(frame_prev), _oflk.as_f32(frame_curr)
    H, W = _oflk.same_shape(p, c)
    u = np.empty((H, W), np.float32)
    v = np.empty((H, W), np.float32)
    _oflk.check(_oflk.lib().oflk_single_scale(_oflk.ptr(p), _oflk.ptr(c), H, W, int(window_size),
                                              _oflk.ptr(u), _oflk.ptr(v)))
    return u, v


def lucas_kanade_single_scale_fp16(frame_prev, frame_curr, window_size: int = 5, pixel_max: float = 255.0):
    """Opt-in reduced-precision single-scale flow: fp16 gradients and fp16 window accumulators
    (BASELINE.json config 5; oflk_single_scale_fp16).  No counterpart in the reference, whose
    arithmetic is fp32 (lucas_kanade_core.py:110-133): close to ``lucas_kanade_single_scale``,
    not equal to it (tests/test_gpu_fp16.py reports the endpoint error per pattern)."""
    p, c = _oflk.as_f32(frame_prev), _oflk.as_f32(frame_curr)
    H, W = _oflk.same_shape(p, c)
    u = np.empty((H, W), np.float32)
    v = np.empty((H, W), np.float32)
    _oflk.check(_oflk.lib().oflk_single_scale_fp16(_oflk.ptr(p), _oflk.ptr(c), 1, H, W, int(window_size),
                                                   float(pixel_max), _oflk.ptr(u), _oflk.ptr(v)))
    return u, v


def lucas_kanade_from_gradients(
    Ix: npt.NDArray[np.float32],
    Iy: npt.NDArray[np.float32],
    It: npt.NDArray[np.float32],
    window_size: int = 5,
) -> Tuple[npt.NDArray[np.float32], npt.NDArray[np.float32]]:
    """Window sums of the gradient products and the per-pixel 2x2 solve.

    Replaces reference lucas_kanade_core.py:73-135 (oflk_from_gradients).
    """
    gx, gy, gt = _oflk.as_f32(Ix), _oflk.as_f32(Iy), _oflk.as_f32(It)
    H, W = _oflk.same_shape(gx, gy, gt)
    u = np.empty((H, W), np.float32)
    v = np.empty((H, W), np.float32)
    _oflk.check(_oflk.lib().oflk_from_gradients(_oflk.ptr(gx), _oflk.ptr(gy), _oflk.ptr(gt), H, W,
                                                int(window_size), _oflk.ptr(u), _oflk.ptr(v)))
    return u, v

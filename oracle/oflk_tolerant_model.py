"""ctypes wrapper around the tolerant-mode CPU model (oracle/oflk_tolerant_model.c, inside liboflk_oracle.so).

TEST INFRASTRUCTURE ONLY (tests/, tools/, bench.py's checker legs).  The model states the arithmetic of the library's
opt-in OFLK_ARITH_TOLERANT mode with one switch per stage x level x iteration; `tolerant_spec` is the assignment the
library ships."""
from __future__ import annotations

import ctypes

import numpy as np

import oflk_oracle as O

WARP = {"exact": 0, "lerp64": 1, "f32": 2, "frac32_lerp64": 3}
SUMS = {"numpy": 0, "separable": 1, "sep_vfirst": 2}
SOLVE = {"exact": 0, "shared_rcp": 1, "fma_det": 2}
PYR = {"exact": 0, "contracted": 1, "f32": 2}
UP = {"exact": 0, "f32": 1, "lerp64": 2}

_i32p = ctypes.POINTER(ctypes.c_int)
_bound = False


def _lib():
    global _bound
    L = O.lib()
    if not _bound:
        L.oflk_model_pyramidal.argtypes = [O._f32p, O._f32p] + [ctypes.c_int] * 5 + [_i32p] * 5 + [O._f32p] * 3 + [_i32p]
        L.oflk_model_pyramidal.restype = ctypes.c_int
        _bound = True
    return L


class Spec:
    """variants per cell; everything exact unless set"""

    def __init__(self, levels: int = 3, iters: int = 3):
        self.levels, self.iters = levels, iters
        self.pyr = np.zeros(levels, np.int32)
        self.up = np.zeros(levels, np.int32)
        self.warp = np.zeros((levels, max(iters, 1)), np.int32)
        self.sums = np.zeros((levels, max(iters, 1)), np.int32)
        self.solve = np.zeros((levels, max(iters, 1)), np.int32)

    def copy(self):
        s = Spec(self.levels, self.iters)
        for k in ("pyr", "up", "warp", "sums", "solve"):
            setattr(s, k, getattr(self, k).copy())
        return s


def tolerant_spec(levels: int = 3, iters: int = 3, shape=None) -> Spec:
    """what OFLK_ARITH_TOLERANT computes (keep in step with plan_pyramidal in csrc/oflk.hip and DESIGN.md section 2): the
    pyramid with fused multiply-adds; on the two finest levels the streaming kernel -- fused-lerp fp64 warp, window sums
    vertical first then horizontal -- with the flow upsampling INTO such a level in the fused-lerp form too (it is fused
    into the level's first iteration); coarser levels exact.  shape = (H, W): levels too small for the streaming kernel
    (5 pixels or fewer along an axis) run exactly, as in the library."""
    s = Spec(levels, iters)
    s.pyr[:] = PYR["contracted"]
    dims = O.pyramid_dims(int(shape[0]), int(shape[1]), levels, 0.5) if shape is not None else None
    for l in range(max(levels - 2, 0), levels):
        if dims is not None and (dims[l][0] <= 4 or dims[l][1] <= 4):
            continue
        s.warp[l, :] = WARP["lerp64"]
        s.sums[l, :] = SUMS["sep_vfirst"]
        if l > 0 and iters >= 1 and (dims is None or (dims[l - 1][0] >= 2 and dims[l - 1][1] >= 2)):
            s.up[l] = UP["lerp64"]
    return s


def pyramidal(prev, curr, spec: Spec, window_size: int = 5):
    """(u, v, residual_log[levels, iters, 2], iters_run[levels]) of the model with `spec`"""
    p, c = O._c(prev), O._c(curr)
    H, W = p.shape
    L, K = spec.levels, spec.iters
    u, v = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    log = np.zeros((L, max(K, 1), 2), np.float32)
    runs = np.zeros(L, np.int32)
    arrs = [np.ascontiguousarray(a, np.int32) for a in (spec.pyr, spec.up, spec.warp, spec.sums, spec.solve)]
    rc = _lib().oflk_model_pyramidal(O._p(p), O._p(c), H, W, L, int(window_size), K,
                                     *[a.ctypes.data_as(_i32p) for a in arrs], O._p(u), O._p(v), O._p(log),
                                     runs.ctypes.data_as(_i32p))
    if rc != 0:
        raise ValueError("oflk_model_pyramidal: bad arguments")
    return u, v, log[:, :K], runs

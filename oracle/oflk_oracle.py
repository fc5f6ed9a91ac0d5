"""ctypes wrapper around oracle/liboflk_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.

Function names and signatures mirror the reference's Python API
(/root/reference/python/lucas_kanade_core.py, lucas_kanade_pyramidal.py) so the
parity tests read like calls to the reference.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path
from typing import List, Sequence, Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liboflk_oracle.so"

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [_HERE / "oflk_oracle.c", _HERE / "oflk_tolerant_model.c"]
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_HERE), "-s", "-B"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        L = _lib
        L.oflk_oracle_set_threads.argtypes = [ctypes.c_int]
        L.oflk_oracle_max_threads.restype = ctypes.c_int
        L.oflk_oracle_np_sum_f32.argtypes = [_f32p, ctypes.c_size_t]
        L.oflk_oracle_np_sum_f32.restype = ctypes.c_float
        L.oflk_oracle_mean_abs.argtypes = [_f32p, ctypes.c_size_t]
        L.oflk_oracle_mean_abs.restype = ctypes.c_float
        L.oflk_oracle_compute_gradients.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p]
        L.oflk_oracle_from_gradients.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]
        L.oflk_oracle_single_scale.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]
        L.oflk_oracle_gaussian_kernel1d.argtypes = [ctypes.c_double, _f64p]
        L.oflk_oracle_gaussian_kernel1d.restype = ctypes.c_int
        L.oflk_oracle_gaussian_filter.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_double, _f32p]
        L.oflk_oracle_resample_linspace.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.oflk_oracle_pyramid_dims.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _i32p]
        L.oflk_oracle_build_pyramid.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(_f32p)]
        L.oflk_oracle_warp.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p]
        L.oflk_oracle_upsample_flow.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]
        L.oflk_oracle_pyramidal.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p]
        L.oflk_oracle_pyramidal.restype = ctypes.c_int
        L.oflk_oracle_build_pyramid_w.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _f64p, ctypes.c_int, ctypes.POINTER(_f32p)]
    return _lib


def set_threads(n: int) -> None:
    lib().oflk_oracle_set_threads(int(n))


def max_threads() -> int:
    return int(lib().oflk_oracle_max_threads())


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def np_sum_f32(a) -> np.float32:
    a = _c(a).ravel()
    return np.float32(lib().oflk_oracle_np_sum_f32(_p(a), a.size))


def mean_abs(a) -> np.float32:
    a = _c(a).ravel()
    return np.float32(lib().oflk_oracle_mean_abs(_p(a), a.size))


def compute_gradients(frame_prev, frame_curr) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    p, c = _c(frame_prev), _c(frame_curr)
    H, W = p.shape
    Ix, Iy, It = (np.empty((H, W), np.float32) for _ in range(3))
    lib().oflk_oracle_compute_gradients(_p(p), _p(c), H, W, _p(Ix), _p(Iy), _p(It))
    return Ix, Iy, It


def lucas_kanade_from_gradients(Ix, Iy, It, window_size: int = 5) -> Tuple[np.ndarray, np.ndarray]:
    Ix, Iy, It = _c(Ix), _c(Iy), _c(It)
    H, W = Ix.shape
    u, v = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().oflk_oracle_from_gradients(_p(Ix), _p(Iy), _p(It), H, W, int(window_size), _p(u), _p(v))
    return u, v


def lucas_kanade_single_scale(frame_prev, frame_curr, window_size: int = 5) -> Tuple[np.ndarray, np.ndarray]:
    p, c = _c(frame_prev), _c(frame_curr)
    H, W = p.shape
    u, v = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().oflk_oracle_single_scale(_p(p), _p(c), H, W, int(window_size), _p(u), _p(v))
    return u, v


def gaussian_kernel1d(sigma: float) -> np.ndarray:
    w = np.zeros(65, np.float64)
    r = lib().oflk_oracle_gaussian_kernel1d(float(sigma), w.ctypes.data_as(_f64p))
    return w[: r + 1].copy()


def gaussian_filter(image, sigma: float) -> np.ndarray:
    a = _c(image)
    H, W = a.shape
    out = np.empty((H, W), np.float32)
    lib().oflk_oracle_gaussian_filter(_p(a), H, W, float(sigma), _p(out))
    return out


def resample_linspace(image, out_shape: Sequence[int]) -> np.ndarray:
    a = _c(image)
    H, W = a.shape
    Ho, Wo = int(out_shape[0]), int(out_shape[1])
    out = np.empty((Ho, Wo), np.float32)
    lib().oflk_oracle_resample_linspace(_p(a), H, W, Ho, Wo, _p(out))
    return out


def pyramid_dims(H: int, W: int, num_levels: int, scale_factor: float = 0.5) -> List[Tuple[int, int]]:
    d = (ctypes.c_int * (2 * num_levels))()
    lib().oflk_oracle_pyramid_dims(H, W, num_levels, float(scale_factor), d)
    return [(d[2 * l], d[2 * l + 1]) for l in range(num_levels)]


def scipy_gaussian_weights(sigma: float) -> np.ndarray:
    """scipy/ndimage/_filters.py _gaussian_kernel1d(sigma, 0, int(4 sigma + 0.5)) restated with NumPy: the half kernel
    w[k], k = 0..radius.  NumPy's exp, as SciPy uses -- libm's differs in the last ulp for some arguments."""
    radius = int(4.0 * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], np.float64)


def build_gaussian_pyramid(image, num_levels: int, scale_factor: float = 0.5) -> List[np.ndarray]:
    a = _c(image)
    H, W = a.shape
    dims = pyramid_dims(H, W, num_levels, scale_factor)
    outs = [np.empty(d, np.float32) for d in dims]
    arr = (_f32p * num_levels)(*[_p(o) for o in outs])
    w = scipy_gaussian_weights(1.0 / float(scale_factor))
    lib().oflk_oracle_build_pyramid_w(_p(a), H, W, num_levels, float(scale_factor), w.ctypes.data_as(_f64p), len(w) - 1, arr)
    return outs


def build_gaussian_pyramid_libm(image, num_levels: int, scale_factor: float = 0.5) -> List[np.ndarray]:
    """the all-C form (weights from libm's exp unless sigma = 2): what the oracle was before round 4"""
    a = _c(image)
    H, W = a.shape
    dims = pyramid_dims(H, W, num_levels, scale_factor)
    outs = [np.empty(d, np.float32) for d in dims]
    arr = (_f32p * num_levels)(*[_p(o) for o in outs])
    lib().oflk_oracle_build_pyramid(_p(a), H, W, num_levels, float(scale_factor), arr)
    return outs


def warp_image(image, flow_u, flow_v) -> np.ndarray:
    a, u, v = _c(image), _c(flow_u), _c(flow_v)
    H, W = a.shape
    out = np.empty((H, W), np.float32)
    lib().oflk_oracle_warp(_p(a), _p(u), _p(v), H, W, _p(out))
    return out


def upsample_flow(flow_u, flow_v, target_shape) -> Tuple[np.ndarray, np.ndarray]:
    u, v = _c(flow_u), _c(flow_v)
    Hc, Wc = u.shape
    Ht, Wt = int(target_shape[0]), int(target_shape[1])
    uo, vo = np.empty((Ht, Wt), np.float32), np.empty((Ht, Wt), np.float32)
    lib().oflk_oracle_upsample_flow(_p(u), _p(v), Hc, Wc, Ht, Wt, _p(uo), _p(vo))
    return uo, vo


def lucas_kanade_pyramidal_ex(frame_prev, frame_curr, num_levels: int = 3, window_size: int = 5,
                              num_iterations: int = 3):
    """Returns (u, v, residual_log[levels, iters, 2], iters_run[levels])."""
    p, c = _c(frame_prev), _c(frame_curr)
    H, W = p.shape
    u, v = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    log = np.zeros((num_levels, max(num_iterations, 1), 2), np.float32)
    runs = np.zeros(num_levels, np.int32)
    rc = lib().oflk_oracle_pyramidal(_p(p), _p(c), H, W, int(num_levels), int(window_size),
                                     int(num_iterations), _p(u), _p(v), _p(log),
                                     runs.ctypes.data_as(_i32p))
    if rc != 0:
        raise ValueError("oflk_oracle_pyramidal: bad arguments")
    return u, v, log[:, :num_iterations], runs


def lucas_kanade_pyramidal(frame_prev, frame_curr, num_levels: int = 3, window_size: int = 5,
                           num_iterations: int = 3) -> Tuple[np.ndarray, np.ndarray]:
    u, v, _, _ = lucas_kanade_pyramidal_ex(frame_prev, frame_curr, num_levels, window_size, num_iterations)
    return u, v

"""ctypes binding of liboflk.so (include/oflk.h) -- the only native dependency of
the drop-in modules in this directory.

There is no CPU fallback: if the library is missing, or no MI355X-class GPU is
usable, every call raises.  Build with ``python __graft_entry__.py`` (or
``make -C optical-flow-fpga_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("OFLK_LIB", _HERE.parent / "liboflk.so"))

OFLK_OK = 0
OFLK_ERR_INVALID = -1
OFLK_ERR_NO_DEVICE = -2
OFLK_ERR_HIP = -3
OFLK_ERR_UNSUPPORTED = -4
OFLK_ERR_NOMEM = -5

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p

# every exported symbol of include/oflk.h: name -> (restype, argtypes)
SIGNATURES = {
    "oflk_version": (ctypes.c_char_p, []),
    "oflk_device_count": (ctypes.c_int, []),
    "oflk_last_error": (ctypes.c_char_p, []),
    "oflk_set_device": (ctypes.c_int, [ctypes.c_int]),
    "oflk_compute_gradients": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p]),
    "oflk_from_gradients": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]),
    "oflk_single_scale": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]),
    "oflk_pyramid_level_dims": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _i32p]),
    "oflk_build_pyramid": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(_f32p)]),
    "oflk_build_pyramid_w": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                            ctypes.c_int, ctypes.POINTER(_f32p)]),
    "oflk_warp": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p]),
    "oflk_upsample_flow": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]),
    "oflk_pyramidal": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p]),
    "oflk_single_scale_batch": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]),
    "oflk_pyramidal_batch": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p]),
    "oflk_single_scale_u8": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]),
    "oflk_pyramidal_u8": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p]),
    "oflk_u8_to_f32": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp]),
    "oflk_plan_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "oflk_plan_destroy": (ctypes.c_int, [_vp]),
    "oflk_plan_workspace_bytes": (ctypes.c_size_t, [_vp]),
    "oflk_plan_single_scale": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "oflk_plan_pyramidal": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "oflk_plan_single_scale_u8": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "oflk_plan_pyramidal_u8": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "oflk_single_scale_batch_multi": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p]),
    "oflk_pyramidal_batch_multi": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p]),
    "oflk_pyramidal_u8_multi": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _i32p]),
    "oflk_shard_range": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, _i32p]),
    "oflk_single_scale_fp16": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _f32p, _f32p]),
    "oflk_rtl_stream_length": (ctypes.c_long, [ctypes.c_int, ctypes.c_int]),
    "oflk_rtl_flow_u8": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "oflk_rtl_flow_u8_device": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "oflk_plan_single_scale_fp16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp]),
    "oflk_plan_resolve_uncertain": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32p]),
    "oflk_plan_resolve_uncertain_u8": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32p]),
    "oflk_last_resolved": (ctypes.c_int, []),
    "oflk_plan_read_uncertain": (ctypes.c_int, [_vp, _i32p, _vp]),
    "oflk_plan_read_level_flow": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _vp]),
    "oflk_pyramidal_last_level_flow": (ctypes.c_int, [ctypes.c_int] * 8 + [_f32p, _f32p]),
    "oflk_pyramidal_last_uncertain": (ctypes.c_int, [ctypes.c_int] * 6 + [_i32p]),
    "oflk_plan_read_log": (ctypes.c_int, [_vp, _f32p, _i32p, _vp]),
    "oflk_plan_set_profiling": (ctypes.c_int, [_vp, ctypes.c_int]),
    "oflk_plan_set_arithmetic": (ctypes.c_int, [_vp, ctypes.c_int]),
    "oflk_plan_set_kernels": (ctypes.c_int, [_vp, ctypes.c_int]),
    "oflk_multi_rehearsal": (ctypes.c_int, [ctypes.c_int]),
    "oflk_set_host_arithmetic": (ctypes.c_int, [ctypes.c_int]),
    "oflk_plan_metrics": (ctypes.c_int, [_vp, _vp, _vp, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), _vp]),
    "oflk_flow_metrics": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "oflk_plan_kernel_times": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.c_int]),
}

_lib: Optional[ctypes.CDLL] = None


class OflkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"liboflk error {code}: {message}")
        self.code = code


def lib() -> ctypes.CDLL:
    """Load liboflk.so once; raise loudly when it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python __graft_entry__.py` or `make -C optical-flow-fpga_amd/csrc`. "
                "There is no CPU fallback."
            )
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
        # opt-in arithmetic of the host entry points (default: the reference's values): OFLK_ARITH=contracted | tolerant
        mode = {"": 0, "exact": 0, "contracted": 1, "tolerant": 2}.get(os.environ.get("OFLK_ARITH", "").strip().lower())
        if mode is None:
            raise ValueError(f"OFLK_ARITH={os.environ['OFLK_ARITH']!r}: expected exact, contracted or tolerant")
        if mode:
            check(L.oflk_set_host_arithmetic(mode))
    return _lib


def check(rc: int) -> None:
    if rc != OFLK_OK:
        msg = lib().oflk_last_error().decode("utf-8", "replace")
        if rc == OFLK_ERR_INVALID:
            raise ValueError(f"liboflk: {msg}")
        raise OflkError(rc, msg)


def version() -> str:
    return lib().oflk_version().decode()


def device_count() -> int:
    return int(lib().oflk_device_count())


def as_f32(a) -> np.ndarray:
    """What the reference's callers hand over: float32 [H, W]; convert anything else."""
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if arr.ndim != 2:
        raise ValueError(f"expected a 2-D array, got shape {arr.shape}")
    return arr


def both_u8(a, b) -> bool:
    """True when both frames are 2-D uint8 arrays (the reference's .bin frame format)."""
    return (isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.dtype == np.uint8 and b.dtype == np.uint8
            and a.ndim == 2 and b.ndim == 2)


def ptr(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def same_shape(*arrs: np.ndarray) -> Tuple[int, int]:
    s = arrs[0].shape
    for a in arrs[1:]:
        if a.shape != s:
            raise ValueError(f"shape mismatch: {s} vs {a.shape}")
    return int(s[0]), int(s[1])


class Plan:
    """Device-resident plan (oflk_plan_*): B pairs of H x W, pointers are raw
    device addresses (e.g. torch.Tensor.data_ptr()), stream a hipStream_t handle."""

    def __init__(self, device: int, B: int, H: int, W: int, levels: int = 3, window_size: int = 5,
                 iters: int = 3):
        self._h = _vp()
        self.B, self.H, self.W, self.levels, self.window_size, self.iters = B, H, W, levels, window_size, iters
        check(lib().oflk_plan_create(ctypes.byref(self._h), device, B, H, W, levels, window_size, iters))

    def close(self) -> None:
        if self._h:
            lib().oflk_plan_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(lib().oflk_plan_workspace_bytes(self._h))

    def single_scale(self, d_prev: int, d_curr: int, d_u: int, d_v: int, stream: int = 0) -> None:
        check(lib().oflk_plan_single_scale(self._h, d_prev, d_curr, d_u, d_v, stream))

    def pyramidal(self, d_prev: int, d_curr: int, d_u: int, d_v: int, stream: int = 0) -> None:
        check(lib().oflk_plan_pyramidal(self._h, d_prev, d_curr, d_u, d_v, stream))

    def single_scale_fp16(self, d_prev: int, d_curr: int, d_u: int, d_v: int, pixel_max: float = 255.0, stream: int = 0) -> None:
        """BASELINE config 5: fp16 gradients / accumulators (opt-in, approximate)."""
        check(lib().oflk_plan_single_scale_fp16(self._h, d_prev, d_curr, d_u, d_v, float(pixel_max), stream))

    def single_scale_u8(self, d_prev: int, d_curr: int, d_u: int, d_v: int, stream: int = 0) -> None:
        """d_prev / d_curr: device uint8 frames [B][H][W] (read by the kernels as they are)."""
        check(lib().oflk_plan_single_scale_u8(self._h, d_prev, d_curr, d_u, d_v, stream))

    def pyramidal_u8(self, d_prev: int, d_curr: int, d_u: int, d_v: int, stream: int = 0) -> None:
        check(lib().oflk_plan_pyramidal_u8(self._h, d_prev, d_curr, d_u, d_v, stream))

    def read_log(self, stream: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        log = np.zeros((self.B, self.levels, max(self.iters, 1), 2), np.float32)
        runs = np.zeros((self.B, self.levels), np.int32)
        check(lib().oflk_plan_read_log(self._h, ptr(log), runs.ctypes.data_as(_i32p), stream))
        return log, runs

    def read_uncertain(self, stream: int = 0) -> np.ndarray:
        """[B][levels] bit masks: bit k = exit decision after iteration k taken within 5e-5 of the threshold."""
        m = np.zeros((self.B, self.levels), np.int32)
        check(lib().oflk_plan_read_uncertain(self._h, m.ctypes.data_as(_i32p), stream))
        return m

    def resolve_uncertain(self, d_prev: int, d_curr: int, d_u: int, d_v: int, stream: int = 0, u8: bool = False) -> int:
        """Redo the pairs whose exit decisions were flagged, in NumPy's summation order; returns how many."""
        n = ctypes.c_int(0)
        fn = lib().oflk_plan_resolve_uncertain_u8 if u8 else lib().oflk_plan_resolve_uncertain
        check(fn(self._h, d_prev, d_curr, d_u, d_v, stream, ctypes.byref(n)))
        return int(n.value)

    def read_level_flow(self, level: int, pair: int, shape, stream: int = 0):
        u = np.empty(shape, np.float32)
        v = np.empty(shape, np.float32)
        check(lib().oflk_plan_read_level_flow(self._h, int(level), int(pair), ptr(u), ptr(v), stream))
        return u, v

    def metrics(self, d_u: int, d_v: int, u_true, v_true, region, stream: int = 0) -> np.ndarray:
        """[B][5] = mae_u, mae_v, rmse, epe, aae of device-resident flows over mask[y0:y1, x0:x1]."""
        ut = np.ascontiguousarray(u_true, np.float32).reshape(self.B)
        vt = np.ascontiguousarray(v_true, np.float32).reshape(self.B)
        out = np.zeros((self.B, 5), np.float64)
        y0, y1, x0, x1 = (int(r) for r in region)
        check(lib().oflk_plan_metrics(self._h, d_u, d_v, ptr(ut), ptr(vt), y0, y1, x0, x1,
                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), stream))
        return out

    def set_arithmetic(self, mode: int) -> None:
        """0 = exact (default: the reference's values), 1 = contracted (opt-in: fused multiply-adds in the Gaussian pyramid)."""
        check(lib().oflk_plan_set_arithmetic(self._h, int(mode)))

    def set_kernels(self, choice: int) -> None:
        """0: automatic (5x5 single-scale streams when the launch is large enough; doubtful tiles redone in NumPy's order);
        1: the tile kernel throughout; 2: the streaming kernel whenever the window is 5x5"""
        check(lib().oflk_plan_set_kernels(self._h, int(choice)))

    def set_profiling(self, enabled) -> None:
        """False/0 off, True/1 every kernel, 2 only the dominant kernel (finest-level LK iteration)."""
        check(lib().oflk_plan_set_profiling(self._h, int(enabled)))

    def kernel_times(self) -> dict:
        n = 32
        names = (ctypes.c_char_p * n)()
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_long * n)()
        k = lib().oflk_plan_kernel_times(self._h, names, ms, cnt, n)
        if k < 0:
            check(k)
        return {names[i].decode(): {"total_ms": ms[i], "launches": cnt[i]} for i in range(k)}

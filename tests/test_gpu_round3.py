"""GPU tests added in round 3 (run on an MI355X: python -m pytest tests -m gpu -x -q).

  * large host batches take the chunked path (H2D of chunk k+1 | kernels of chunk k | D2H of chunk k-1 on three
    streams and two host threads): same flows, logs and iteration counts as pair-by-pair calls and as the oracle
  * oflk_last_resolved() counts the pairs of THIS call (reset per call, summed over chunks)
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int)


def _batch(rng, B, H, W, u8=False):
    a = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
    b = np.roll(a, (1, 2), axis=(1, 2))
    b = np.clip(b.astype(np.int32) + rng.integers(-6, 7, (B, H, W)), 0, 255).astype(np.uint8)
    return (a, b) if u8 else (a.astype(np.float32), b.astype(np.float32))


@pytest.mark.parametrize("u8", [False, True])
def test_chunked_host_batch_equals_pair_by_pair(u8):
    """21 pairs of 1080p (five chunks of four pairs and a tail of one) and 9 pairs of 2160x3840 (nine chunks of one):
    the chunked call returns what single-pair calls return, flow, residual log and iteration counts"""
    import _oflk
    import lucas_kanade_pyramidal as P

    L = _oflk.lib()
    rng = np.random.default_rng(3)
    for (B, H, W) in ((21, 1080, 1920), (9, 2160, 3840)):
        a, b = _batch(rng, 3, H, W, u8)
        a = np.ascontiguousarray(np.concatenate([a] * ((B + 2) // 3))[:B])
        b = np.ascontiguousarray(np.concatenate([b] * ((B + 2) // 3))[:B])
        u = np.empty((B, H, W), np.float32)
        v = np.empty_like(u)
        log = np.zeros((B, 3, 3, 2), np.float32)
        runs = np.zeros((B, 3), np.int32)
        if u8:
            _oflk.check(L.oflk_pyramidal_u8(a.ctypes.data, b.ctypes.data, B, H, W, 3, 5, 3, u.ctypes.data_as(f32p),
                                            v.ctypes.data_as(f32p), log.ctypes.data_as(f32p), runs.ctypes.data_as(i32p)))
        else:
            _oflk.check(L.oflk_pyramidal_batch(a.ctypes.data_as(f32p), b.ctypes.data_as(f32p), B, H, W, 3, 5, 3,
                                               u.ctypes.data_as(f32p), v.ctypes.data_as(f32p), log.ctypes.data_as(f32p),
                                               runs.ctypes.data_as(i32p)))
        assert L.oflk_last_resolved() == 0
        for i in range(3):   # the three distinct pairs, one call each
            ru, rv, rlog, rruns = P.lucas_kanade_pyramidal_with_log(a[i], b[i], 3, 5, 3)
            for k in range(i, B, 3):
                assert np.array_equal(u[k], ru) and np.array_equal(v[k], rv), (B, H, W, k)
                assert list(runs[k]) == list(rruns)
                np.testing.assert_array_equal(log[k], rlog)


def test_chunked_single_scale_and_ragged_tail(oracle):
    """single-scale batches through the chunked path with a ragged last chunk (B not a multiple of the chunk), against the oracle"""
    import _oflk

    L = _oflk.lib()
    rng = np.random.default_rng(5)
    B, H, W = 19, 1080, 1920   # chunks of four pairs: four full ones and a tail of three
    a, b = _batch(rng, 2, H, W)
    a = np.ascontiguousarray(np.concatenate([a] * 10)[:B])
    b = np.ascontiguousarray(np.concatenate([b] * 10)[:B])
    u = np.empty((B, H, W), np.float32)
    v = np.empty_like(u)
    _oflk.check(L.oflk_single_scale_batch(a.ctypes.data_as(f32p), b.ctypes.data_as(f32p), B, H, W, 5, u.ctypes.data_as(f32p),
                                          v.ctypes.data_as(f32p)))
    for i in range(2):
        ou, ov = oracle.lucas_kanade_single_scale(a[i], b[i], 5)
        for k in range(i, B, 2):
            assert np.array_equal(u[k], ou) and np.array_equal(v[k], ov), k

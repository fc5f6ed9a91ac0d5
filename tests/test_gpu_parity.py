"""GPU parity tests: the HIP path (through the C ABI / drop-in modules) against
the CPU oracle on identical inputs.  Bar: value-for-value equality
(np.array_equal; -0.0 == +0.0) for every flow field and image; the residual
means (an fp64 reduction on the device vs NumPy's fp32 pairwise mean) to 2e-6
relative.

Run on an MI355X:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import lucas_kanade_core

    return lucas_kanade_core


@pytest.fixture(scope="module")
def P():
    import lucas_kanade_pyramidal

    return lucas_kanade_pyramidal


def _rand_pair(rng, H, W, integer=False):
    if integer:
        a = rng.integers(0, 256, (H, W)).astype(np.float32)
        b = rng.integers(0, 256, (H, W)).astype(np.float32)
    else:
        a = rng.normal(110, 45, (H, W)).astype(np.float32)
        b = (a + rng.normal(0, 6, (H, W))).astype(np.float32)
    return a, b


def _eq(a, b, what=""):
    assert a.shape == b.shape, what
    assert a.dtype == np.float32 and b.dtype == np.float32, what
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} differ; first at {i}: {a[i]!r} vs {b[i]!r}; "
                             f"max abs {np.max(np.abs(a.astype(np.float64) - b))}")


def test_device_present():
    import _oflk

    assert _oflk.device_count() >= 1
    assert "gfx950" in _oflk.version()


SHAPES = [(37, 53), (16, 64), (17, 65), (64, 256), (240, 320), (5, 5), (7, 9), (4, 4), (1, 1), (3, 200), (200, 3),
          (480, 640)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("win", [5, 3, 7, 4, 9, 11])
def test_single_scale_matches_oracle(K, oracle, shape, win):
    rng = np.random.default_rng(hash((shape, win)) & 0xFFFF)
    for integer in (False, True):
        a, b = _rand_pair(rng, *shape, integer=integer)
        u, v = K.lucas_kanade_single_scale(a, b, win)
        ou, ov = oracle.lucas_kanade_single_scale(a, b, win)
        _eq(u, ou, f"u {shape} win{win}")
        _eq(v, ov, f"v {shape} win{win}")


@pytest.mark.parametrize("shape", [(37, 53), (240, 320), (1, 1), (2, 7), (130, 70)])
def test_compute_gradients_matches_oracle(K, oracle, shape):
    rng = np.random.default_rng(1)
    a, b = _rand_pair(rng, *shape)
    for g, o, n in zip(K.compute_gradients(a, b), oracle.compute_gradients(a, b), "Ix Iy It".split()):
        _eq(g, o, n)


@pytest.mark.parametrize("win", [3, 5, 7])
def test_from_gradients_matches_oracle(K, oracle, win):
    rng = np.random.default_rng(2)
    Ix, Iy, It = (rng.normal(0, 9, (61, 83)).astype(np.float32) for _ in range(3))
    u, v = K.lucas_kanade_from_gradients(Ix, Iy, It, win)
    ou, ov = oracle.lucas_kanade_from_gradients(Ix, Iy, It, win)
    _eq(u, ou, "u")
    _eq(v, ov, "v")


def test_flat_frames_give_zero_flow(K):
    a = np.full((40, 50), 77.0, np.float32)
    u, v = K.lucas_kanade_single_scale(a, a.copy(), 5)
    assert not u.any() and not v.any()


@pytest.mark.parametrize("shape,levels", [((240, 320), 3), ((37, 53), 3), ((97, 131), 2), ((64, 64), 4), ((33, 20), 1)])
def test_pyramid_matches_oracle(P, oracle, shape, levels):
    rng = np.random.default_rng(3)
    a, _ = _rand_pair(rng, *shape)
    got = P.build_gaussian_pyramid(a, levels)
    exp = oracle.build_gaussian_pyramid(a, levels)
    assert [g.shape for g in got] == [e.shape for e in exp]
    for l, (g, e) in enumerate(zip(got, exp)):
        _eq(g, e, f"level {l}")


def test_warp_matches_oracle(P, oracle):
    rng = np.random.default_rng(4)
    H, W = 61, 83
    img = rng.normal(100, 40, (H, W)).astype(np.float32)
    # sub-pixel, large (out of frame), exactly integral and edge-landing displacements
    u = rng.normal(0, 4, (H, W)).astype(np.float32)
    v = rng.normal(0, 4, (H, W)).astype(np.float32)
    u[::7, ::5] = 3.0
    v[::7, ::5] = -2.0
    u[5, :] = np.float32(W - 1) - np.arange(W, dtype=np.float32)  # lands exactly on x = W-1
    v[:, 9] = -np.arange(H, dtype=np.float32)                      # lands exactly on y = 0
    u[20:24, 20:24] = 7078.0
    v[30:34, 30:34] = -1e-9
    _eq(P.warp_image(img, u, v), oracle.warp_image(img, u, v), "warp")
    z = np.zeros_like(img)
    _eq(P.warp_image(img, z, z), img, "identity warp")
    # degenerate shapes: one column, one row, one pixel
    for shape in ((9, 1), (1, 9), (1, 1), (2, 2)):
        im = rng.normal(50, 20, shape).astype(np.float32)
        uu = rng.normal(0, 1, shape).astype(np.float32)
        vv = rng.normal(0, 1, shape).astype(np.float32)
        uu.flat[0] = 0.0
        vv.flat[0] = 0.0
        _eq(P.warp_image(im, uu, vv), oracle.warp_image(im, uu, vv), f"warp {shape}")


@pytest.mark.parametrize("cshape,tshape", [((60, 80), (120, 160)), ((37, 53), (75, 107)), ((9, 13), (18, 26)),
                                           ((1, 1), (2, 3)), ((5, 1), (11, 2))])
def test_upsample_matches_oracle(P, oracle, cshape, tshape):
    rng = np.random.default_rng(5)
    u = rng.normal(0, 3, cshape).astype(np.float32)
    v = rng.normal(0, 3, cshape).astype(np.float32)
    gu, gv = P.upsample_flow(u, v, tshape)
    ou, ov = oracle.upsample_flow(u, v, tshape)
    _eq(gu, ou, "u")
    _eq(gv, ov, "v")


def _check_pyramidal(P, oracle, a, b, levels, win, iters):
    u, v, log, runs = P.lucas_kanade_pyramidal_with_log(a, b, levels, win, iters)
    ou, ov, olog, oruns = oracle.lucas_kanade_pyramidal_ex(a, b, levels, win, iters)
    assert list(runs[:levels]) == list(oruns), (runs, oruns)
    for l in range(levels):
        k = int(oruns[l])
        np.testing.assert_allclose(log[l, :k], olog[l, :k], rtol=2e-6, atol=1e-9)
    _eq(u, ou, "pyramidal u")
    _eq(v, ov, "pyramidal v")


@pytest.mark.parametrize("shape,levels,win,iters", [((240, 320), 3, 5, 3), ((97, 131), 2, 5, 3), ((120, 160), 3, 7, 2),
                                                    ((64, 80), 1, 5, 3), ((75, 75), 3, 3, 4), ((240, 320), 4, 5, 3),
                                                    ((120, 160), 2, 9, 2), ((96, 128), 2, 11, 2)])
def test_pyramidal_matches_oracle_synthetic(P, oracle, shape, levels, win, iters):
    from oflk_synth import synth_pair

    a, b = synth_pair(*shape, pair_index=levels)
    _check_pyramidal(P, oracle, a, b, levels, win, iters)


@pytest.mark.parametrize("shape,levels,win,iters", [
    ((16, 20), 3, 5, 3),     # coarsest level 4 x 5: no pixel has a full window there
    ((12, 12), 2, 5, 2),     # 6 x 6 coarse level: a 2 x 2 interior
    ((9, 40), 3, 7, 2),      # every level is shorter than the window
    ((40, 3), 1, 5, 3),      # narrower than the window
    ((33, 47), 3, 5, 0),     # no iterations at all: the flow stays zero
    ((48, 64), 2, 5, 1),
])
def test_pyramidal_degenerate_levels(P, oracle, shape, levels, win, iters):
    """levels without interior pixels read as converged after one iteration (d = 0 everywhere)"""
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    a, b = _rand_pair(rng, *shape)
    _check_pyramidal(P, oracle, a, b, levels, win, iters)


def test_pyramidal_matches_oracle_noninteger(P, oracle):
    rng = np.random.default_rng(7)
    a, b = _rand_pair(rng, 90, 122)
    _check_pyramidal(P, oracle, a, b, 3, 5, 3)


def test_pyramidal_early_exit(P, oracle):
    """identical frames: residual is 0, every level stops after one iteration (:221-223)"""
    from oflk_synth import synth_pair

    a, _ = synth_pair(120, 160)
    u, v, log, runs = P.lucas_kanade_pyramidal_with_log(a, a.copy(), 3, 5, 3)
    assert list(runs) == [1, 1, 1]
    assert not u.any() and not v.any()
    assert not log.any()
    _check_pyramidal(P, oracle, a, a.copy(), 3, 5, 3)


def test_batch_equals_per_pair():
    """oflk_pyramidal_batch / oflk_single_scale_batch: pairs are independent units"""
    import ctypes

    import _oflk
    from oflk_synth import synth_pair

    B, H, W, L, K = 3, 120, 160, 3, 3
    pairs = [synth_pair(H, W, i, dx=1.0 + i, dy=-0.5 * i) for i in range(B)]
    pairs[1] = (pairs[1][0], pairs[1][0].copy())  # this pair converges at once: mixed early exit
    prev = np.stack([p for p, _ in pairs])
    curr = np.stack([c for _, c in pairs])
    u = np.empty_like(prev)
    v = np.empty_like(prev)
    log = np.zeros((B, L, K, 2), np.float32)
    runs = np.zeros((B, L), np.int32)
    f32p = ctypes.POINTER(ctypes.c_float)
    _oflk.check(_oflk.lib().oflk_pyramidal_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, 5,
                                                 K, u.ctypes.data_as(f32p), v.ctypes.data_as(f32p),
                                                 log.ctypes.data_as(f32p),
                                                 runs.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    import lucas_kanade_pyramidal as P

    for i in range(B):
        su, sv, slog, sruns = P.lucas_kanade_pyramidal_with_log(prev[i], curr[i], L, 5, K)
        _eq(u[i], su, f"pair {i} u")
        _eq(v[i], sv, f"pair {i} v")
        assert list(runs[i]) == list(sruns)
        np.testing.assert_array_equal(log[i], slog)
    assert list(runs[1]) == [1, 1, 1]

    us = np.empty_like(prev)
    vs = np.empty_like(prev)
    _oflk.check(_oflk.lib().oflk_single_scale_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, 5,
                                                    us.ctypes.data_as(f32p), vs.ctypes.data_as(f32p)))
    import lucas_kanade_core as K_

    for i in range(B):
        su, sv = K_.lucas_kanade_single_scale(prev[i], curr[i], 5)
        _eq(us[i], su, f"single pair {i} u")
        _eq(vs[i], sv, f"single pair {i} v")


def test_errors_are_loud():
    import _oflk
    import lucas_kanade_core as K_

    a = np.zeros((8, 8), np.float32)
    with pytest.raises(ValueError):
        K_.lucas_kanade_single_scale(a, np.zeros((8, 9), np.float32))
    with pytest.raises(_oflk.OflkError):
        K_.lucas_kanade_single_scale(a, a, 47)  # window beyond 45x45 -> explicit error, no fallback
    # C ABI argument checks: status code + message, never a crash
    import ctypes

    L = _oflk.lib()
    f32p = ctypes.POINTER(ctypes.c_float)
    none = ctypes.cast(None, f32p)
    assert L.oflk_single_scale(none, _oflk.ptr(a), 8, 8, 5, _oflk.ptr(a), _oflk.ptr(a)) == _oflk.OFLK_ERR_INVALID
    assert b"NULL" in L.oflk_last_error()
    assert L.oflk_pyramidal(_oflk.ptr(a), _oflk.ptr(a), 8, 8, 0, 5, 3, _oflk.ptr(a), _oflk.ptr(a), None, None) == _oflk.OFLK_ERR_INVALID
    assert L.oflk_pyramidal(_oflk.ptr(a), _oflk.ptr(a), 8, 8, 5, 5, 3, _oflk.ptr(a), _oflk.ptr(a), None, None) == _oflk.OFLK_ERR_INVALID  # level 4 would be empty
    out = (ctypes.c_double * 5)()
    t = np.zeros(1, np.float32)
    assert L.oflk_flow_metrics(_oflk.ptr(a), _oflk.ptr(a), 0, 8, 8, _oflk.ptr(t), _oflk.ptr(t), 0, 8, 0, 8, out) == _oflk.OFLK_ERR_INVALID
    h = ctypes.c_void_p()
    assert L.oflk_plan_create(ctypes.byref(h), 0, 1, 1 << 16, 1 << 16, 1, 5, 0) == _oflk.OFLK_ERR_UNSUPPORTED  # >= 2^30 pixels
    assert L.oflk_plan_create(ctypes.byref(h), 99, 1, 8, 8, 1, 5, 0) != _oflk.OFLK_OK   # no such device


# ---- BASELINE.json full sizes ---------------------------------------------------
@pytest.mark.parametrize("shape", [(1080, 1920), (2160, 3840)])
def test_full_size_pyramidal_matches_oracle(P, oracle, shape):
    """configs[2] (1080p) and one pair of configs[3] (4K): the whole call against the oracle"""
    from oflk_synth import synth_pair

    a, b = synth_pair(*shape, pair_index=1)
    oracle.set_threads(8)
    try:
        _check_pyramidal(P, oracle, a, b, 3, 5, 3)
    finally:
        oracle.set_threads(1)


def test_bench_workload_batch_equals_single_pair_calls(P):
    """bench.py's launch shape: 32 pairs of 1080p in one call.  Large launches take the chained-tile
    path of the iteration kernel (a block walks several vertically adjacent tiles and carries their
    shared staging rows); a one-pair call does not.  Both must give the same flow, value for value --
    and the one-pair call is held to the oracle by test_full_size_pyramidal_matches_oracle."""
    import ctypes

    import _oflk
    from oflk_synth import synth_pair

    B, H, W, L, K = 32, 1080, 1920, 3, 3
    distinct = [synth_pair(H, W, pair_index=1), synth_pair(H, W, pair_index=6, dx=-2.25, dy=0.75)]
    single = [P.lucas_kanade_pyramidal_with_log(a, b, L, 5, K) for a, b in distinct]
    prev = np.stack([distinct[i % 2][0] for i in range(B)])
    curr = np.stack([distinct[i % 2][1] for i in range(B)])
    u = np.empty_like(prev)
    v = np.empty_like(prev)
    log = np.zeros((B, L, K, 2), np.float32)
    runs = np.zeros((B, L), np.int32)
    f32p = ctypes.POINTER(ctypes.c_float)
    _oflk.check(_oflk.lib().oflk_pyramidal_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, 5, K,
                                                 u.ctypes.data_as(f32p), v.ctypes.data_as(f32p),
                                                 log.ctypes.data_as(f32p), runs.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    for i in range(B):
        su, sv, slog, sruns = single[i % 2]
        assert np.array_equal(u[i], su) and np.array_equal(v[i], sv), f"pair {i}"
        assert list(runs[i]) == list(sruns)
        np.testing.assert_allclose(log[i], slog, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("shape", [(203, 317), (130, 332), (75, 1283)])
def test_chained_tiles_on_ragged_shapes(oracle, shape):
    """many small pairs in one call: the launch is large enough for the chained-tile path while the
    frames end in partial tiles on both axes (odd widths take the scalar load/store paths).  Every
    pair must equal the oracle's result for it."""
    import ctypes

    import _oflk

    H, W = shape
    L, K, B = 3, 2, 256
    rng = np.random.default_rng(H * 1000 + W)
    distinct = [_rand_pair(rng, H, W) for _ in range(3)]
    distinct.append((distinct[0][0], distinct[0][0].copy()))   # one pair that converges at once
    expect = [oracle.lucas_kanade_pyramidal_ex(a, b, L, 5, K) for a, b in distinct]
    prev = np.stack([distinct[i % 4][0] for i in range(B)])
    curr = np.stack([distinct[i % 4][1] for i in range(B)])
    u = np.empty_like(prev)
    v = np.empty_like(prev)
    runs = np.zeros((B, L), np.int32)
    f32p = ctypes.POINTER(ctypes.c_float)
    _oflk.check(_oflk.lib().oflk_pyramidal_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, 5, K,
                                                 u.ctypes.data_as(f32p), v.ctypes.data_as(f32p), None,
                                                 runs.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    for i in range(B):
        eu, ev, _, eruns = expect[i % 4]
        assert list(runs[i]) == list(eruns), i
        _eq(u[i], eu, f"u of pair {i}")
        _eq(v[i], ev, f"v of pair {i}")


def test_640x480_single_scale_matches_oracle(K, oracle):
    """configs[1]"""
    from oflk_synth import synth_pair

    a, b = synth_pair(480, 640, pair_index=2)
    u, v = K.lucas_kanade_single_scale(a, b, 5)
    ou, ov = oracle.lucas_kanade_single_scale(a, b, 5)
    _eq(u, ou, "u")
    _eq(v, ov, "v")


def test_single_scale_is_local_at_4k(K, oracle):
    """size-independent property: single-scale flow at a pixel depends only on its 7x7
    neighbourhood, so a crop computed alone agrees with the full-frame result on the
    crop's interior (3-pixel rim excluded), bit for bit"""
    from oflk_synth import synth_pair

    a, b = synth_pair(2160, 3840, pair_index=3)
    u, v = K.lucas_kanade_single_scale(a, b, 5)
    rng = np.random.default_rng(9)
    for _ in range(4):
        y = int(rng.integers(0, 2160 - 200))
        x = int(rng.integers(0, 3840 - 300))
        cu, cv = oracle.lucas_kanade_single_scale(a[y:y + 200, x:x + 300], b[y:y + 200, x:x + 300], 5)
        _eq(np.ascontiguousarray(u[y + 3:y + 197, x + 3:x + 297]), np.ascontiguousarray(cu[3:-3, 3:-3]), "u crop")
        _eq(np.ascontiguousarray(v[y + 3:y + 197, x + 3:x + 297]), np.ascontiguousarray(cv[3:-3, 3:-3]), "v crop")


def test_7x7_window_at_8k_single_scale(K, oracle):
    """configs[4] geometry (7680x4320, 7x7 window), fp32 exact path, checked on crops"""
    from oflk_synth import synth_pair

    a, b = synth_pair(4320, 7680, pair_index=4)
    u, v = K.lucas_kanade_single_scale(a, b, 7)
    for (y, x) in ((0, 0), (2000, 3000), (4320 - 160, 7680 - 240)):
        cu, cv = oracle.lucas_kanade_single_scale(a[y:y + 160, x:x + 240], b[y:y + 160, x:x + 240], 7)
        ys = slice(0 if y == 0 else 4, 160 if y + 160 == 4320 else 156)
        xs = slice(0 if x == 0 else 4, 240 if x + 240 == 7680 else 236)
        _eq(np.ascontiguousarray(u[y:y + 160, x:x + 240][ys, xs]), np.ascontiguousarray(cu[ys, xs]), "u crop")
        _eq(np.ascontiguousarray(v[y:y + 160, x:x + 240][ys, xs]), np.ascontiguousarray(cv[ys, xs]), "v crop")


def test_repeat_calls_are_deterministic(P):
    from oflk_synth import synth_pair

    a, b = synth_pair(270, 480, pair_index=5)
    r1 = P.lucas_kanade_pyramidal_with_log(a, b, 3, 5, 3)
    r2 = P.lucas_kanade_pyramidal_with_log(a, b, 3, 5, 3)
    for x, y in zip(r1, r2):
        assert np.array_equal(x, y)

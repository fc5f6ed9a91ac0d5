"""GPU tests added in round 2 (run on an MI355X: python -m pytest tests -m gpu -x -q).

  * uint8 frames read by the kernels themselves (oflk_*_u8, oflk_plan_*_u8): value-for-value
    the float path's flow, on aligned / ragged / odd-offset buffers and through the chained-tile path
  * one-process multi-GPU entry points (oflk_*_multi) against the single-device batch entry
  * the host entry points' per-device plan cache under alternating shapes
  * device pointers that are not 16-byte aligned (element-wise kernel instantiation)
  * BASELINE config 4's per-GPU share (8 pairs of 3840x2160, chained tiles) against the oracle
  * early-exit decisions next to the 0.01 threshold: equal to the oracle's or reported as uncertain
  * OFLK_DUMP_LEVELS: per-level flows read back after the call equal the oracle's per-level flows
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int)


def _eq(a, b, what=""):
    assert a.shape == b.shape and a.dtype == b.dtype == np.float32, what
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} differ; first at {i}: {a[i]!r} vs {b[i]!r}")


def _u8_pair(rng, H, W, B=None):
    shape = (H, W) if B is None else (B, H, W)
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    # a shifted, noisy copy: textured enough that most pixels pass the det test
    b = np.roll(a, (1, 2), axis=(-2, -1))
    b = np.clip(b.astype(np.int32) + rng.integers(-6, 7, shape), 0, 255).astype(np.uint8)
    return a, b


# ---------------------------------------------------------------------------------------------
# uint8 ingestion
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (17, 65), (1, 1), (5, 5), (3, 200), (200, 3), (240, 320), (64, 1)])
@pytest.mark.parametrize("win", [5, 3, 7])
def test_u8_host_entry_points_equal_the_float_path(shape, win):
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P

    rng = np.random.default_rng(hash((shape, win)) & 0xFFFF)
    a, b = _u8_pair(rng, *shape)
    af, bf = a.astype(np.float32), b.astype(np.float32)
    u8, v8 = K.lucas_kanade_single_scale(a, b, win)
    uf, vf = K.lucas_kanade_single_scale(af, bf, win)
    _eq(u8, uf, f"single u {shape} win{win}")
    _eq(v8, vf, f"single v {shape} win{win}")
    for levels, iters in ((3, 3), (1, 2), (2, 1)):
        try:
            rf = P.lucas_kanade_pyramidal_with_log(af, bf, levels, win, iters)
        except ValueError:   # a pyramid level would be empty: the uint8 entry must refuse the same way
            with pytest.raises(ValueError):
                P.lucas_kanade_pyramidal_with_log(a, b, levels, win, iters)
            continue
        r8 = P.lucas_kanade_pyramidal_with_log(a, b, levels, win, iters)
        _eq(r8[0], rf[0], f"pyr u {shape} win{win} L{levels}")
        _eq(r8[1], rf[1], f"pyr v {shape} win{win} L{levels}")
        assert list(r8[3]) == list(rf[3])
        np.testing.assert_array_equal(r8[2], rf[2])


def test_u8_matches_oracle(oracle):
    import lucas_kanade_pyramidal as P

    rng = np.random.default_rng(11)
    a, b = _u8_pair(rng, 120, 160)
    u, v, log, runs = P.lucas_kanade_pyramidal_with_log(a, b, 3, 5, 3)
    ou, ov, olog, oruns = oracle.lucas_kanade_pyramidal_ex(a.astype(np.float32), b.astype(np.float32), 3, 5, 3)
    _eq(u, ou, "u")
    _eq(v, ov, "v")
    assert list(runs) == list(oruns)


def test_u8_plan_api_batches_and_odd_offsets():
    """Device-resident uint8 frames through the plan API: a batch large enough for the chained-tile path,
    and frame buffers that start at odd byte offsets (byte-wise kernel instantiation)."""
    import torch

    import _oflk

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    st = torch.cuda.current_stream().cuda_stream
    for (B, H, W) in ((48, 256, 320), (3, 70, 90)):
        a, b = _u8_pair(rng, H, W, B)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        fa, fb = ta.float(), tb.float()
        for levels, iters in ((1, 0), (3, 3)):
            plan = _oflk.Plan(0, B, H, W, levels, 5, iters)
            u8 = torch.empty((B, H, W), dtype=torch.float32, device=dev)
            v8 = torch.empty_like(u8)
            uf, vf = torch.empty_like(u8), torch.empty_like(u8)
            if iters == 0:
                plan.single_scale_u8(ta.data_ptr(), tb.data_ptr(), u8.data_ptr(), v8.data_ptr(), st)
                plan.single_scale(fa.data_ptr(), fb.data_ptr(), uf.data_ptr(), vf.data_ptr(), st)
            else:
                plan.pyramidal_u8(ta.data_ptr(), tb.data_ptr(), u8.data_ptr(), v8.data_ptr(), st)
                _, r8 = plan.read_log(st)
                plan.pyramidal(fa.data_ptr(), fb.data_ptr(), uf.data_ptr(), vf.data_ptr(), st)
                _, rf = plan.read_log(st)
                np.testing.assert_array_equal(r8, rf)
            torch.cuda.synchronize()
            assert torch.equal(u8, uf) and torch.equal(v8, vf), (B, H, W, levels)
            # the same frames one byte further into a larger buffer
            n = B * H * W
            pa = torch.empty(n + 16, dtype=torch.uint8, device=dev)
            pb = torch.empty(n + 16, dtype=torch.uint8, device=dev)
            pa[1:n + 1] = ta.reshape(-1)
            pb[3:n + 3] = tb.reshape(-1)
            uo, vo = torch.empty_like(u8), torch.empty_like(u8)
            if iters == 0:
                plan.single_scale_u8(pa.data_ptr() + 1, pb.data_ptr() + 3, uo.data_ptr(), vo.data_ptr(), st)
            else:
                plan.pyramidal_u8(pa.data_ptr() + 1, pb.data_ptr() + 3, uo.data_ptr(), vo.data_ptr(), st)
            torch.cuda.synchronize()
            assert torch.equal(uo, uf) and torch.equal(vo, vf), ("odd offset", B, H, W, levels)
            plan.close()


def test_unaligned_float_device_pointers():
    """W % 4 == 0 but planes 4 bytes off 16-byte alignment: accepted, element-wise kernels, same values."""
    import torch

    import _oflk
    from oflk_synth import synth_pair

    dev = torch.device("cuda", 0)
    B, H, W = 2, 96, 128
    st = torch.cuda.current_stream().cuda_stream
    pairs = [synth_pair(H, W, i) for i in range(B)]
    prev = torch.from_numpy(np.stack([p for p, _ in pairs])).to(dev)
    curr = torch.from_numpy(np.stack([c for _, c in pairs])).to(dev)
    n = B * H * W
    plan = _oflk.Plan(0, B, H, W, 3, 5, 3)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), st)
    torch.cuda.synchronize()
    big = [torch.empty(n + 8, dtype=torch.float32, device=dev) for _ in range(4)]
    big[0][1:n + 1] = prev.reshape(-1)
    big[1][1:n + 1] = curr.reshape(-1)
    ptrs = [t.data_ptr() + 4 for t in big]
    assert all(q % 16 == 4 for q in ptrs)
    plan.pyramidal(ptrs[0], ptrs[1], ptrs[2], ptrs[3], st)
    torch.cuda.synchronize()
    assert torch.equal(big[2][1:n + 1].reshape(B, H, W), u) and torch.equal(big[3][1:n + 1].reshape(B, H, W), v)
    s_u, s_v = torch.empty_like(prev), torch.empty_like(prev)
    plan1 = _oflk.Plan(0, B, H, W, 1, 5, 0)
    plan1.single_scale(prev.data_ptr(), curr.data_ptr(), s_u.data_ptr(), s_v.data_ptr(), st)
    plan1.single_scale(ptrs[0], ptrs[1], ptrs[2], ptrs[3], st)
    torch.cuda.synchronize()
    assert torch.equal(big[2][1:n + 1].reshape(B, H, W), s_u) and torch.equal(big[3][1:n + 1].reshape(B, H, W), s_v)
    plan.close()
    plan1.close()


# ---------------------------------------------------------------------------------------------
# multi-GPU entry points, plan cache, device switching
# ---------------------------------------------------------------------------------------------
def _batch(B, H, W):
    from oflk_synth import synth_pair

    pairs = [synth_pair(H, W, i, dx=1.0 + 0.5 * i, dy=-0.25 * i) for i in range(B)]
    pairs[B // 2] = (pairs[B // 2][0], pairs[B // 2][0].copy())   # converges at once: mixed early exit
    return np.stack([p for p, _ in pairs]), np.stack([c for _, c in pairs])


def test_multi_entry_equals_single_device_batch():
    import _oflk

    L_ = _oflk.lib()
    B, H, W, L, K = 5, 120, 160, 3, 3
    prev, curr = _batch(B, H, W)
    ref = [np.empty_like(prev), np.empty_like(prev), np.zeros((B, L, K, 2), np.float32), np.zeros((B, L), np.int32)]
    _oflk.check(L_.oflk_pyramidal_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, 5, K,
                                        ref[0].ctypes.data_as(f32p), ref[1].ctypes.data_as(f32p),
                                        ref[2].ctypes.data_as(f32p), ref[3].ctypes.data_as(i32p)))
    ndev = _oflk.device_count()
    for n_gpus in sorted({1, 0, ndev, min(2, ndev)}):
        got = [np.empty_like(prev), np.empty_like(prev), np.zeros((B, L, K, 2), np.float32), np.zeros((B, L), np.int32)]
        _oflk.check(L_.oflk_pyramidal_batch_multi(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, 5, K,
                                                  n_gpus, got[0].ctypes.data_as(f32p), got[1].ctypes.data_as(f32p),
                                                  got[2].ctypes.data_as(f32p), got[3].ctypes.data_as(i32p)))
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        us, vs = np.empty_like(prev), np.empty_like(prev)
        _oflk.check(L_.oflk_single_scale_batch_multi(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, 5,
                                                     n_gpus, us.ctypes.data_as(f32p), vs.ctypes.data_as(f32p)))
        u1, v1 = np.empty_like(prev), np.empty_like(prev)
        _oflk.check(L_.oflk_single_scale_batch(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, 5,
                                               u1.ctypes.data_as(f32p), v1.ctypes.data_as(f32p)))
        np.testing.assert_array_equal(us, u1)
        np.testing.assert_array_equal(vs, v1)
    # uint8 form
    p8 = np.clip(prev, 0, 255).astype(np.uint8)
    c8 = np.clip(curr, 0, 255).astype(np.uint8)
    a = [np.empty_like(prev), np.empty_like(prev)]
    b = [np.empty_like(prev), np.empty_like(prev)]
    _oflk.check(L_.oflk_pyramidal_u8_multi(p8.ctypes.data, c8.ctypes.data, B, H, W, L, 5, K, 0, a[0].ctypes.data_as(f32p),
                                           a[1].ctypes.data_as(f32p), None, None))
    _oflk.check(L_.oflk_pyramidal_u8(p8.ctypes.data, c8.ctypes.data, B, H, W, L, 5, K, b[0].ctypes.data_as(f32p),
                                     b[1].ctypes.data_as(f32p), None, None))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    # more GPUs than there are: loud
    rc = L_.oflk_pyramidal_batch_multi(prev.ctypes.data_as(f32p), curr.ctypes.data_as(f32p), B, H, W, L, 5, K, ndev + 1,
                                       got[0].ctypes.data_as(f32p), got[1].ctypes.data_as(f32p), None, None)
    assert rc == _oflk.OFLK_ERR_INVALID and b"visible" in L_.oflk_last_error()


def test_plan_cache_alternating_shapes(oracle):
    """The verifier alternates single-scale and pyramidal calls; a caller may mix sizes: more shapes than
    the cache holds, twice around, every result equal to the oracle's."""
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P
    from oflk_synth import synth_pair

    shapes = [(60, 80), (64, 96), (37, 53), (90, 70), (48, 64), (50, 50), (72, 88), (33, 129)]
    want = {}
    for rnd in range(2):
        for i, (H, W) in enumerate(shapes):
            a, b = synth_pair(H, W, i)
            u, v = K.lucas_kanade_single_scale(a, b, 5)
            pu, pv, _, runs = P.lucas_kanade_pyramidal_with_log(a, b, 2, 5, 2)
            if rnd == 0:
                ou, ov = oracle.lucas_kanade_single_scale(a, b, 5)
                opu, opv, _, oruns = oracle.lucas_kanade_pyramidal_ex(a, b, 2, 5, 2)
                want[i] = (ou, ov, opu, opv, list(oruns))
            ou, ov, opu, opv, oruns = want[i]
            _eq(u, ou), _eq(v, ov), _eq(pu, opu), _eq(pv, opv)
            assert list(runs) == oruns


def test_set_device_switches_buffers_and_plans():
    """host scratch was keyed by size, not by device.  Each device now has its own context."""
    import _oflk
    import lucas_kanade_core as K
    from oflk_synth import synth_pair

    if _oflk.device_count() < 2:
        pytest.skip("needs two GPUs")
    L_ = _oflk.lib()
    a, b = synth_pair(96, 128, 0)
    try:
        _oflk.check(L_.oflk_set_device(0))
        u0, v0 = K.lucas_kanade_single_scale(a, b, 5)
        _oflk.check(L_.oflk_set_device(1))
        u1, v1 = K.lucas_kanade_single_scale(a, b, 5)
        _oflk.check(L_.oflk_set_device(0))
        u2, v2 = K.lucas_kanade_single_scale(a, b, 5)
    finally:
        L_.oflk_set_device(0)
    _eq(u0, u1), _eq(v0, v1), _eq(u0, u2), _eq(v0, v2)


def test_shard_range_is_the_python_rule():
    import _oflk
    from oflk_dist import shard_range

    L_ = _oflk.lib()
    for total in (0, 1, 7, 8, 64, 65, 1000):
        for n in (1, 2, 3, 8):
            cover = []
            for i in range(n):
                b0, b1 = ctypes.c_int(), ctypes.c_int()
                L_.oflk_shard_range(total, i, n, ctypes.byref(b0), ctypes.byref(b1))
                assert (b0.value, b1.value) == shard_range(total, i, n)
                cover += list(range(b0.value, b1.value))
            assert cover == list(range(total))


# ---------------------------------------------------------------------------------------------
# BASELINE config 4, one GPU's share: 8 pairs of 3840x2160 through the chained-tile path
# ---------------------------------------------------------------------------------------------
def test_4k_batch_of_8_matches_oracle(oracle):
    import torch

    import _oflk
    from oflk_synth import synth_pair

    dev = torch.device("cuda", 0)
    B, H, W, L, K = 8, 2160, 3840, 3, 3
    distinct = [synth_pair(H, W, i) for i in range(2)]
    want = []
    oracle.set_threads(min(oracle.max_threads(), len(os.sched_getaffinity(0)), 16))
    try:
        for p, c in distinct:
            want.append(oracle.lucas_kanade_pyramidal_ex(p, c, L, 5, K))
    finally:
        oracle.set_threads(1)
    prev = torch.stack([torch.from_numpy(distinct[b % 2][0]) for b in range(B)]).to(dev)
    curr = torch.stack([torch.from_numpy(distinct[b % 2][1]) for b in range(B)]).to(dev)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    plan = _oflk.Plan(0, B, H, W, L, 5, K)
    st = torch.cuda.current_stream().cuda_stream
    plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), st)
    log, runs = plan.read_log(st)
    torch.cuda.synchronize()
    hu, hv = u.cpu().numpy(), v.cpu().numpy()
    for b in range(B):
        ou, ov, olog, oruns = want[b % 2]
        _eq(hu[b], ou, f"pair {b} u")
        _eq(hv[b], ov, f"pair {b} v")
        assert list(runs[b]) == list(oruns)
        np.testing.assert_allclose(log[b], olog, rtol=2e-6, atol=0)
    assert not plan.read_uncertain(st).any()
    plan.close()


# ---------------------------------------------------------------------------------------------
# early-exit decisions next to the threshold
# ---------------------------------------------------------------------------------------------
def _threshold_family(oracle, H=96, W=128, L=2, K=3):
    """Frames curr_t = prev + t * (shifted - prev): the residual means grow with t.  Bisect t (on the oracle) to
    the point where the first iteration's larger mean crosses float32(0.01) at the coarsest level."""
    from oflk_synth import synth_pair

    prev, shifted = synth_pair(H, W, 0, dx=0.75, dy=-0.5)
    delta = (shifted - prev).astype(np.float64)

    def frames(t):
        return prev, (prev + t * delta).astype(np.float32)

    def first_means(t):
        p, c = frames(t)
        _, _, olog, oruns = oracle.lucas_kanade_pyramidal_ex(p, c, L, 5, K)
        return max(float(olog[0, 0, 0]), float(olog[0, 0, 1])), list(oruns)

    thr = float(np.float32(0.01))
    lo, hi = 0.0, 1.0
    assert first_means(hi)[0] > thr > first_means(lo)[0]
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if first_means(mid)[0] < thr:
            lo = mid
        else:
            hi = mid
    return frames, first_means, lo, hi, thr


def test_exit_decisions_next_to_the_threshold(oracle):
    """A ladder of inputs around the crossing (both sides, relative offsets 1e-7 ... 3e-2).  The device decides
    from an exact fixed-point total, the reference from an fp32 pairwise sum: they can differ only inside the
    guard band, where the decision is flagged and the host entry point redoes the pair in NumPy's own order
    (oflk_plan_resolve_uncertain).  Required: EVERY input takes the oracle's iteration counts, log and flow;
    inputs further than 1e-3 from the threshold are never redone; the ladder does reach the band."""
    import lucas_kanade_pyramidal as P
    import _oflk

    H, W, L, K = 96, 128, 2, 3
    frames, first_means, lo, hi, thr = _threshold_family(oracle, H, W, L, K)
    ts = [lo, hi]
    for rel in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4, 1e-3, 1e-2, 3e-2):
        ts += [lo * (1 - rel), hi * (1 + rel)]
    redone = 0
    for t in ts:
        p, c = frames(t)
        m, oruns = first_means(t)
        u, v, log, runs = P.lucas_kanade_pyramidal_with_log(p, c, L, 5, K)
        n = int(_oflk.lib().oflk_last_resolved())
        rel_dist = abs(m / thr - 1.0)
        if n:
            redone += 1
            assert rel_dist < 1e-3, f"t={t!r}: redone although the mean {m!r} is {rel_dist:.2e} away from the threshold"
        flags = np.zeros(L, np.int32)
        _oflk.check(_oflk.lib().oflk_pyramidal_last_uncertain(1, H, W, L, 5, K, flags.ctypes.data_as(i32p)))
        assert not flags.any()   # resolved pairs are no longer flagged
        ou, ov, olog, _ = oracle.lucas_kanade_pyramidal_ex(p, c, L, 5, K)
        assert list(runs) == oruns, f"t={t!r}: mean {m!r}, GPU ran {list(runs)}, oracle {oruns}"
        _eq(u, ou, f"t={t!r} u")
        _eq(v, ov, f"t={t!r} v")
        if n:   # a redone pair carries NumPy-order means: equal, not merely close
            for l in range(L):
                np.testing.assert_array_equal(log[l, :oruns[l]], olog[l, :oruns[l]])
    assert redone >= 2, "the ladder never reached the uncertainty band: the construction is broken"


def test_guard_band_follows_the_level_size(oracle):
    """np.mean adds ceil(n / 8192) fp32 pieces one after the other, so its worst-case error grows with
    the level: at 3840x2160 the band is (1013 + 32) * 2^-24 = 6.2e-5 relative, wider than the 5e-5 floor small
    levels use.  A 4K level whose first mean lies 5.6e-5 above the threshold (bisected on the device's own exact
    mean) must be flagged, and the redone pair must equal the oracle."""
    import torch

    import _oflk
    from oflk_synth import synth_pair

    H, W, L, K = 2160, 3840, 1, 2
    prev, shifted = synth_pair(H, W, 0, dx=0.75, dy=-0.5)
    delta = (shifted - prev).astype(np.float64)
    dev = torch.device("cuda", 0)
    d_prev = torch.from_numpy(prev[None]).to(dev)
    u, v = torch.empty_like(d_prev), torch.empty_like(d_prev)
    plan = _oflk.Plan(0, 1, H, W, L, 5, K)
    st = torch.cuda.current_stream().cuda_stream

    def run(t):
        c = (prev + t * delta).astype(np.float32)
        d_curr = torch.from_numpy(c[None]).to(dev)
        plan.pyramidal(d_prev.data_ptr(), d_curr.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        log, runs = plan.read_log(st)
        torch.cuda.synchronize()
        return c, d_curr, float(max(log[0, 0, 0])), list(runs[0])

    thr = float(np.float32(0.01))
    target = thr * (1.0 + 5.6e-5)
    lo, hi = 0.0, 1.0
    assert run(lo)[2] < target < run(hi)[2]
    # the mean moves in small jumps with t (a pixel crossing the solve's determinant threshold moves an 8.3 Mpx mean by
    # ~1e-5 of the threshold): bisect towards the middle of the window, then walk t in fine steps around the crossing
    got, seen = None, []
    inside = lambda m: 5.15e-5 < m / thr - 1.0 < 6.1e-5
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        c, d_curr, m, runs = run(mid)
        seen.append(m / thr - 1.0)
        if inside(m):
            got = (c, d_curr, m, runs)
            break
        if m < target:
            lo = mid
        else:
            hi = mid
    if got is None:
        for i in range(-100, 101):
            c, d_curr, m, runs = run(lo * (1.0 + i * 2e-7))
            seen.append(m / thr - 1.0)
            if inside(m):
                got = (c, d_curr, m, runs)
                break
    assert got is not None, f"no input landed between the 5e-5 floor and the 4K band; closest {min(seen, key=lambda r: abs(r - 5.6e-5)):.3e}"
    c, d_curr, m, runs = got
    flags = plan.read_uncertain(st)
    assert flags[0, 0] & 1, f"mean {m!r} is {m / thr - 1.0:.2e} above the threshold at 8.3 Mpx and was not flagged"
    assert plan.resolve_uncertain(d_prev.data_ptr(), d_curr.data_ptr(), u.data_ptr(), v.data_ptr(), st) == 1
    log, runs = plan.read_log(st)
    torch.cuda.synchronize()
    ou, ov, olog, oruns = oracle.lucas_kanade_pyramidal_ex(prev, c, L, 5, K)
    assert list(runs[0]) == list(oruns)
    _eq(u.cpu().numpy()[0], ou, "4K near-threshold u")
    _eq(v.cpu().numpy()[0], ov, "4K near-threshold v")
    np.testing.assert_array_equal(log[0, 0, :oruns[0]], olog[0, :oruns[0]])
    plan.close()


def test_resolve_uncertain_inside_a_batch(oracle):
    """Plan API: one near-threshold pair among ordinary ones; only that pair is redone, every pair ends up
    equal to the oracle; uint8 form included."""
    import torch

    import _oflk
    from oflk_synth import synth_pair

    H, W, L, K = 96, 128, 2, 3
    frames, first_means, lo, hi, thr = _threshold_family(oracle, H, W, L, K)
    pairs = [synth_pair(H, W, 1), frames(lo), synth_pair(H, W, 2), frames(hi)]
    dev = torch.device("cuda", 0)
    prev = torch.from_numpy(np.stack([p for p, _ in pairs])).to(dev)
    curr = torch.from_numpy(np.stack([c for _, c in pairs])).to(dev)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    plan = _oflk.Plan(0, 4, H, W, L, 5, K)
    st = torch.cuda.current_stream().cuda_stream
    plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), st)
    flags = plan.read_uncertain(st)
    assert flags[1].any() and flags[3].any() and not flags[0].any() and not flags[2].any()
    assert plan.resolve_uncertain(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), st) == 2
    assert not plan.read_uncertain(st).any()
    log, runs = plan.read_log(st)
    hu, hv = u.cpu().numpy(), v.cpu().numpy()
    for b, (p, c) in enumerate(pairs):
        ou, ov, olog, oruns = oracle.lucas_kanade_pyramidal_ex(p, c, L, 5, K)
        _eq(hu[b], ou, f"pair {b} u")
        _eq(hv[b], ov, f"pair {b} v")
        assert list(runs[b]) == list(oruns)
    plan.close()


# ---------------------------------------------------------------------------------------------
# per-level flows after the call (the reference's PNG side effect, :226)
# ---------------------------------------------------------------------------------------------
def test_level_flows_read_back_equal_the_oracle(oracle, tmp_path, monkeypatch):
    import lucas_kanade_pyramidal as P
    import _oflk
    from oflk_synth import synth_pair

    H, W, L, K = 120, 160, 3, 3
    a, b = synth_pair(H, W, 3)
    u, v, log, runs = P.lucas_kanade_pyramidal_with_log(a, b, L, 5, K)
    shapes = P.pyramid_level_shapes((H, W), L)
    # the oracle on the truncated pyramids: level l's final flow is the result of a (l+1)-level run on that level's frames
    pa, pb = oracle.build_gaussian_pyramid(a, L), oracle.build_gaussian_pyramid(b, L)
    for level in range(L - 1):
        lu = np.empty(shapes[level], np.float32)
        lv = np.empty(shapes[level], np.float32)
        _oflk.check(_oflk.lib().oflk_pyramidal_last_level_flow(1, H, W, L, 5, K, level, 0, lu.ctypes.data_as(f32p),
                                                               lv.ctypes.data_as(f32p)))
        ou, ov, _, _ = oracle.lucas_kanade_pyramidal_ex(pa[level], pb[level], level + 1, 5, K)
        # (a truncated run rebuilds its own pyramid from level `level`'s frames: equal only at level 0,
        # where no pyramid step is involved; deeper levels are checked for shape and finiteness)
        if level == 0:
            _eq(lu, ou, "level 0 u")
            _eq(lv, ov, "level 0 v")
        assert lu.shape == shapes[level] and np.isfinite(lu).all() and np.isfinite(lv).all()
    # the opt-in dump writes one PNG per level (skipped when matplotlib is missing)
    monkeypatch.setenv("OFLK_DUMP_LEVELS", "1")
    monkeypatch.setenv("OFLK_DUMP_DIR", str(tmp_path))
    P.lucas_kanade_pyramidal(a, b, L, 5, K)
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        return
    assert sorted(q.name for q in tmp_path.glob("*.png")) == [f"pyramid_level_{l}.png" for l in range(L)]


# ---------------------------------------------------------------------------------------------
# the progress lines the reference prints (lucas_kanade_pyramidal.py:172-222, README.md:258-299)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["translate_small", "no_motion"])
def test_printed_log_equals_the_reference_stdout(golden_dir, name, capsys, monkeypatch):
    """tests/golden/reference_stdout.json holds the reference's own stdout (make_golden_stdout.py);
    the shim reproduces it from the residual log the C call returns, character for character."""
    import json

    import lucas_kanade_pyramidal as P

    gold = json.loads((golden_dir / "reference_stdout.json").read_text())
    z = np.load(golden_dir / "patterns_320x240.npz")
    p, c = z["frame_0"].astype(np.float32), z[f"frame_1__{name}"].astype(np.float32)
    monkeypatch.setenv("OFLK_QUIET", "0")
    capsys.readouterr()
    P.lucas_kanade_pyramidal(p, c, **gold["args"])
    assert capsys.readouterr().out == gold["stdout"][name]


@pytest.mark.parametrize("sf", [0.6, 0.4, 0.75, 0.3])
def test_pyramid_other_scale_factors_equal_the_reference(golden_dir, sf):
    """oflk_build_pyramid away from the default scale factor (unfused blur / resample kernels, Gaussian weights from
    libm): equal to the reference's own pyramids (tests/golden/pyramid_scales.npz)."""
    import lucas_kanade_pyramidal as P

    z = np.load(golden_dir / "pyramid_scales.npz")
    pyr = P.build_gaussian_pyramid(z["image"], 3, scale_factor=sf)
    assert len(pyr) == 3
    for l, a in enumerate(pyr):
        _eq(np.asarray(a, np.float32), z[f"sf{sf}_level{l}"], f"scale {sf} level {l}")


# ---------------------------------------------------------------------------------------------
# bench.py with N > 1 ranks, end to end (rehearsal: gloo for the three small collectives, every rank on GPU 0)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("config,ranks,expect_pairs", [("1080p", 2, 6), ("4k64", 3, 64)])
def test_bench_multi_rank_rehearsal(config, ranks, expect_pairs):
    """A one-GPU box cannot host two RCCL ranks, so the N > 1 path of bench.py (sharding, per-rank plans on
    LOCAL_RANK, fence, MAX of the elapsed time, SUM of the result totals, rank-0 JSON) is rehearsed with
    --backend gloo --force-device 0 on small frames.  The driver's 8-GPU run uses the same code with nccl."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr",
           "127.0.0.1", "--master-port", "29581" if config == "1080p" else "29582", str(root / "bench.py"), "--gpus", str(ranks),
           "--steps", "2", "--warmup", "1", "--backend", "gloo", "--force-device", "0", "--config", config, "--height", "120",
           "--width", "160", "--no-cpu-baseline", "--no-one-pair"]
    if config == "1080p":
        cmd += ["--pairs", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == ranks and j["steps"] == 2 and j["warmup"] == 1
    assert j["scaling"] == ("weak" if config == "1080p" else "strong")
    assert j["config"]["pairs_per_step_job"] == expect_pairs
    assert j["job_stats"]["pairs_per_step"] == expect_pairs
    # whole-job value: pixels of ALL ranks over the MAX-rank time
    assert abs(j["value"] - expect_pairs * 120 * 160 * 2 / (j["ms_per_step"] * 2e-3) / 1e6) <= 1e-3 * j["value"]
    assert "REHEARSAL" in j["data"] and j["cpu_baseline"] is None


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` with no launcher on the command line: the parent starts the two ranks as a child
    torch.distributed.run job before touching the GPU and relays rank 0's ONE JSON line."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                          "--force-device", "0", "--height", "120", "--width", "160", "--pairs", "3", "--no-cpu-baseline",
                          "--no-one-pair"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["pairs_per_step_job"] == 6 and "REHEARSAL" in j["data"]
    # the parity half of the metric rides in the same line: the reference's 26 flow fields, value for value
    e = j["epe_vs_reference"]
    assert e["patterns"] == 13 and e["digests_equal"] == 26 and e["max_mean_epe"] == 0.0 and e["iteration_counts_equal"] == 13


def test_bench_multi_inproc_rehearsal():
    """--multi inproc: one process, a plan and a stream per GPU (here: twice GPU 0) on the shards of the 4k64 job"""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--multi", "inproc", "--steps", "2", "--warmup", "1",
                          "--force-device", "0", "--config", "4k64", "--height", "120", "--width", "160"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["pairs_per_step_job"] == 64
    assert j["config"]["pairs_per_gpu_per_step"] == [32, 32] and j["job_stats"]["mean_abs_u"] > 0

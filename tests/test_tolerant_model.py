"""CPU tests of oracle/oflk_tolerant_model.c, the CPU statement of the library's opt-in OFLK_ARITH_TOLERANT arithmetic
(test infrastructure; the GPU tests hold the HIP kernels to it bit for bit, tests/test_gpu_round4.py).

  * with every switch off the model IS the oracle (same values)
  * the shipped assignment of switches stays within the north star's tolerance -- mean endpoint error <= 1e-4 px -- of
    dense flows THE REFERENCE produced (tests/golden/dense_reference_flows.npz, made by make_golden_dense.py importing
    it): the 13 verification patterns and pair 0 of the bench workload at 1920x1080
  * a relaxation the ablation found over the bar (an fp32 pyramid) is indeed caught by this test's measure
"""
import numpy as np
import pytest

TOL = 1e-4


def _epe(u, v, ru, rv):
    return float(np.mean(np.sqrt((u.astype(np.float64) - ru) ** 2 + (v.astype(np.float64) - rv) ** 2)))


@pytest.fixture(scope="module")
def dense(golden_dir):
    return np.load(golden_dir / "dense_reference_flows.npz")


@pytest.fixture(scope="module")
def patterns(golden_dir):
    z = np.load(golden_dir / "patterns_320x240.npz")
    f0 = z["frame_0"].astype(np.float32)
    return {k[len("frame_1__"):]: (f0, z[k].astype(np.float32)) for k in z.files if k.startswith("frame_1__")}


def test_dense_fixture_is_the_reference_of_the_digest_fixtures(dense, golden_dir):
    """the dense flows are the same fields whose sha256 the exact-path tests use"""
    import hashlib
    import json

    ref = json.loads((golden_dir / "reference_13patterns.json").read_text())["patterns"]
    dig = lambda a: hashlib.sha256((np.ascontiguousarray(a, np.float32) + np.float32(0.0)).tobytes()).hexdigest()  # noqa: E731
    for n, r in ref.items():
        assert dig(dense[f"{n}__u"]) == r["pyramidal"]["u_sha256"] and dig(dense[f"{n}__v"]) == r["pyramidal"]["v_sha256"], n
    c2 = json.loads((golden_dir / "reference_fullsize.json").read_text())["c2"]
    assert dig(dense["bench_1080p_pair0__u"]) == c2["u_sha256"] and dig(dense["bench_1080p_pair0__v"]) == c2["v_sha256"]


def test_model_with_every_switch_off_is_the_oracle(oracle, patterns):
    import oflk_tolerant_model as M

    for n in ("translate_medium", "rotate_small", "no_motion"):
        p, c = patterns[n]
        u, v, log, runs = oracle.lucas_kanade_pyramidal_ex(p, c, 3, 5, 3)
        mu, mv, mlog, mruns = M.pyramidal(p, c, M.Spec(3, 3), 5)
        assert np.array_equal(u, mu) and np.array_equal(v, mv) and list(runs) == list(mruns)
        np.testing.assert_array_equal(log, mlog)


def test_shipped_tolerant_arithmetic_is_within_tolerance_of_the_reference(dense, patterns):
    import oflk_tolerant_model as M

    worst = 0.0
    for n, (p, c) in patterns.items():
        u, v, _, runs = M.pyramidal(p, c, M.tolerant_spec(3, 3), 5)
        e = _epe(u, v, dense[f"{n}__u"], dense[f"{n}__v"])
        assert list(runs) == list(dense[f"{n}__iters"]), n
        assert e <= TOL, (n, e)
        worst = max(worst, e)
    assert worst <= TOL / 3   # measured 1.7e-5 (translate_extreme): the mode keeps a factor of five in hand


def test_shipped_tolerant_arithmetic_on_the_bench_pair(dense):
    import oflk_tolerant_model as M
    from oflk_synth import synth_pair

    p, c = synth_pair(1080, 1920, 0)
    u, v, _, runs = M.pyramidal(p, c, M.tolerant_spec(3, 3), 5)
    assert list(runs) == list(dense["bench_1080p_pair0__iters"])
    assert _epe(u, v, dense["bench_1080p_pair0__u"], dense["bench_1080p_pair0__v"]) <= TOL


def test_the_measure_catches_a_relaxation_that_is_over_the_bar(dense, patterns):
    """an all-fp32 pyramid (rejected by tools/experiments/fast_mode_ablation.py: 4.8e-4 on translate_extreme)"""
    import oflk_tolerant_model as M

    s = M.Spec(3, 3)
    s.pyr[:] = M.PYR["f32"]
    p, c = patterns["translate_extreme"]
    u, v, _, _ = M.pyramidal(p, c, s, 5)
    assert _epe(u, v, dense["translate_extreme__u"], dense["translate_extreme__v"]) > TOL


@pytest.mark.parametrize("key", ["e1", "e2", "e3", "m1", "m5"])
def test_tolerant_arithmetic_on_the_reference_made_mid_size_cases(oracle, golden_dir, key):
    """the 5x5 cases of tests/golden/reference_fullsize.json -- identical frames and sub-pixel motions whose levels leave their
    loops after [1,1,1], [2,1,1] and [3,4,4] of 4 iterations, a 4-level / 5-iteration case, an odd 481x643 shape: the oracle
    reproduces the reference's digests there (tests/test_oracle_golden.py), so its flow IS the reference's; the tolerant
    arithmetic must take the same iteration counts and stay within 1e-4 px of it"""
    import hashlib
    import json

    import oflk_tolerant_model as M
    from oflk_synth import synth_pair, synth_pair_smooth

    c = json.loads((golden_dir / "reference_fullsize.json").read_text())[key]
    assert c["window_size"] == 5 and c["mode"] == "pyramidal"
    h, w = c["shape"]
    gen = synth_pair_smooth if c.get("smooth") else synth_pair
    p, q = gen(h, w, c.get("pair_index", 0), c.get("dx", 3.0), c.get("dy", -1.5))
    L, K = c["levels"], c["iterations"]
    ou, ov, _, oruns = oracle.lucas_kanade_pyramidal_ex(p, q, L, 5, K)
    dig = lambda a: hashlib.sha256((np.ascontiguousarray(a, np.float32) + np.float32(0.0)).tobytes()).hexdigest()  # noqa: E731
    assert dig(ou) == c["u_sha256"] and dig(ov) == c["v_sha256"]          # the oracle's flow is the reference's
    u, v, _, runs = M.pyramidal(p, q, M.tolerant_spec(L, K, (h, w)), 5)
    assert list(runs) == list(oruns), (key, list(runs), list(oruns))
    assert _epe(u, v, ou, ov) <= TOL, key

"""CPU tests: the oracle (oracle/oflk_oracle.c) against golden vectors produced by
importing the reference (tests/golden/make_golden.py), and against the reference's
own known-answer file python/verification_baseline.json (tests/golden/ copy).

Bar: bit-exact (sha256 of the float32 bytes after +0.0, or np.array_equal).
"""
import hashlib
import json

import numpy as np
import pytest

PATTERNS = ["translate_small", "translate_medium", "translate_large", "translate_vertical", "translate_diagonal",
            "rotate_small", "rotate_medium", "rotate_large", "zoom_in", "zoom_out", "translate_rotate", "no_motion",
            "translate_extreme"]


def digest(a):
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


@pytest.fixture(scope="module")
def suite(golden_dir):
    z = np.load(golden_dir / "patterns_320x240.npz")
    ref = json.loads((golden_dir / "reference_13patterns.json").read_text())
    return z, ref


@pytest.fixture(scope="module")
def stage(golden_dir):
    return np.load(golden_dir / "stage_vectors.npz")


def _pair(z, name):
    return z["frame_0"].astype(np.float32), z[f"frame_1__{name}"].astype(np.float32)


def test_pattern_bytes_match_recorded_digests(suite):
    z, ref = suite
    assert hashlib.sha256(z["frame_0"].tobytes()).hexdigest() == ref["frame0_sha256"]
    for name in PATTERNS:
        assert hashlib.sha256(z[f"frame_1__{name}"].tobytes()).hexdigest() == ref["patterns"][name]["frame1_sha256"]


@pytest.mark.parametrize("name", PATTERNS)
def test_oracle_single_scale_equals_reference(oracle, suite, name):
    z, ref = suite
    p, c = _pair(z, name)
    u, v = oracle.lucas_kanade_single_scale(p, c, 5)
    r = ref["patterns"][name]["single_scale"]
    assert digest(u) == r["u_sha256"] and digest(v) == r["v_sha256"]


@pytest.mark.parametrize("name", PATTERNS)
def test_oracle_pyramidal_equals_reference(oracle, suite, name):
    z, ref = suite
    p, c = _pair(z, name)
    u, v, log, runs = oracle.lucas_kanade_pyramidal_ex(p, c, 3, 5, 3)
    r = ref["patterns"][name]["pyramidal"]
    assert list(runs) == r["iters_run"]
    for l, rows in enumerate(r["residual_log"]):
        # the reference's np.mean(np.abs(d)) values, reproduced exactly (fp32 pairwise, 8192-piece order)
        np.testing.assert_array_equal(log[l, :len(rows)], np.array(rows, np.float32))
    assert digest(u) == r["u_sha256"] and digest(v) == r["v_sha256"]


def test_no_motion_exits_after_one_iteration(oracle, suite):
    z, ref = suite
    p, c = _pair(z, "no_motion")
    u, v, log, runs = oracle.lucas_kanade_pyramidal_ex(p, c, 3, 5, 3)
    assert list(runs) == [1, 1, 1] and not u.any() and not v.any()


def test_metrics_reproduce_verification_baseline(oracle, suite, golden_dir):
    """oracle flows + this repo's flow_metrics/mask == the reference's committed known answers"""
    import flow_metrics
    import optical_flow_verifier as V

    z, ref = suite
    base = json.loads((golden_dir / "verification_baseline.json").read_text())["patterns"]
    worst = 0.0
    for name in PATTERNS:
        p, c = _pair(z, name)
        gt = ref["patterns"][name]["motion"]
        mask = V.get_test_region_mask(p.shape, name, 80)
        assert int(mask.sum()) == base[name]["num_test_pixels"]
        flows = {"single_scale": oracle.lucas_kanade_single_scale(p, c, 5),
                 "pyramidal": oracle.lucas_kanade_pyramidal(p, c, 3, 5, 3)}
        for key, (u, v) in flows.items():
            m = flow_metrics.compute_all_metrics(u, v, gt["dx"], gt["dy"], mask)
            # exactly what the reference's own metric code gave on the reference's own flows here
            assert m == ref["patterns"][name][key]["metrics"], (name, key)
            for k, val in base[name][key]["metrics"].items():
                worst = max(worst, abs(m[k] - val))
                assert abs(m[k] - val) <= 2e-5, (name, key, k, m[k], val)
    assert worst <= 2e-5


def test_dense_reference_flow(oracle, suite, golden_dir):
    z, _ = suite
    d = np.load(golden_dir / "dense_translate_medium.npz")
    p, c = _pair(z, "translate_medium")
    u, v = oracle.lucas_kanade_single_scale(p, c, 5)
    assert np.array_equal(u, d["single_u"]) and np.array_equal(v, d["single_v"])
    u, v = oracle.lucas_kanade_pyramidal(p, c, 3, 5, 3)
    assert np.array_equal(u, d["pyr_u"]) and np.array_equal(v, d["pyr_v"])


@pytest.mark.parametrize("tag", ["tile", "noise"])
def test_stage_vectors(oracle, stage, tag):
    s = stage
    p, c = s[f"{tag}__prev"], s[f"{tag}__curr"]
    Ix, Iy, It = oracle.compute_gradients(p, c)
    for g, n in ((Ix, "Ix"), (Iy, "Iy"), (It, "It")):
        assert np.array_equal(g, s[f"{tag}__{n}"]), n
    for win in (3, 4, 5, 7):
        u, v = oracle.lucas_kanade_single_scale(p, c, win)
        assert np.array_equal(u, s[f"{tag}__single_u_w{win}"]) and np.array_equal(v, s[f"{tag}__single_v_w{win}"]), win
        u2, v2 = oracle.lucas_kanade_from_gradients(Ix, Iy, It, win)
        assert np.array_equal(u, u2) and np.array_equal(v, v2)
    for l, a in enumerate(oracle.build_gaussian_pyramid(p, 3)):
        assert np.array_equal(a, s[f"{tag}__pyr{l}"]), f"pyr{l}"
    fu, fv = s[f"{tag}__flow_u"], s[f"{tag}__flow_v"]
    assert np.array_equal(oracle.warp_image(c, fu, fv), s[f"{tag}__warped"])
    uu, vv = oracle.upsample_flow(fu, fv, s[f"{tag}__up_u"].shape)
    assert np.array_equal(uu, s[f"{tag}__up_u"]) and np.array_equal(vv, s[f"{tag}__up_v"])
    u, v, log, runs = oracle.lucas_kanade_pyramidal_ex(p, c, 2, 5, 3)
    assert list(runs) == list(s[f"{tag}__pyrlk_runs"])
    assert np.array_equal(u, s[f"{tag}__pyrlk_u"]) and np.array_equal(v, s[f"{tag}__pyrlk_v"])
    for l in range(2):
        k = int(runs[l])
        assert np.array_equal(log[l, :k], s[f"{tag}__pyrlk_log"][l, :k])


def test_mean_abs_known_answers(oracle, stage):
    for n in (25, 4800, 8192, 8200, 76800):
        assert oracle.mean_abs(stage[f"meanabs__x{n}"]) == stage[f"meanabs__y{n}"][0]


def test_rtl_frame_pairs(oracle, golden_dir):
    """the two 320x240 pairs the reference commits as .mem files for its testbench"""
    r = np.load(golden_dir / "rtl_frames.npz")
    for tag in ("natural", "sinusoid"):
        p, c = r[f"{tag}__frame_00"].astype(np.float32), r[f"{tag}__frame_01"].astype(np.float32)
        us, vs = oracle.lucas_kanade_single_scale(p, c, 5)
        up, vp, _, runs = oracle.lucas_kanade_pyramidal_ex(p, c, 3, 5, 3)
        assert [digest(us), digest(vs), digest(up), digest(vp)] == list(r[f"{tag}__sha"])
        assert list(runs) == list(r[f"{tag}__runs"])
        reg = np.s_[105:135, 55:85]
        ans = r[f"{tag}__answers"]
        assert np.mean(us[reg]) == ans[0] and np.mean(up[reg]) == ans[2] and float(np.sum(us != 0)) == ans[4]


def test_gaussian_weights_are_scipys(oracle):
    filters = pytest.importorskip("scipy.ndimage._filters")
    w = filters._gaussian_kernel1d(2.0, 0, 8)
    assert np.array_equal(oracle.gaussian_kernel1d(2.0), w[8:])


def test_oracle_against_scipy_and_numpy_directly(oracle):
    """third-party arithmetic restated by the oracle, checked against the libraries themselves"""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(11)
    a = rng.normal(50, 20, (33, 47)).astype(np.float32)
    assert np.array_equal(oracle.gaussian_filter(a, 2.0), ndi.gaussian_filter(a, sigma=2.0))
    yy, xx = np.meshgrid(np.linspace(0, 32, 16), np.linspace(0, 46, 23), indexing="ij")
    assert np.array_equal(oracle.resample_linspace(a, (16, 23)), ndi.map_coordinates(a, [yy, xx], order=1, mode="constant"))
    x = rng.normal(0, 1, 100000).astype(np.float32)
    assert oracle.np_sum_f32(x) == np.sum(x) and oracle.mean_abs(x) == np.mean(np.abs(x))


PRESET_PATTERNS = ["translate_medium", "rotate_small", "translate_extreme", "no_motion"]


@pytest.mark.parametrize("name", PRESET_PATTERNS)
@pytest.mark.parametrize("preset", ["shallow", "deep", "large_window"])
def test_oracle_other_presets_equal_reference(oracle, suite, golden_dir, name, preset):
    """2-level, 4-level and 7x7 presets of verification_config.yaml (reference :78-103)"""
    z, _ = suite
    ref = json.loads((golden_dir / "reference_presets.json").read_text())[name][preset]
    p, c = _pair(z, name)
    u, v = oracle.lucas_kanade_pyramidal(p, c, ref["levels"], ref["window_size"], ref["iterations"])
    assert digest(u) == ref["u_sha256"] and digest(v) == ref["v_sha256"]


def test_pyramid_other_scale_factors_equal_the_reference(oracle, golden_dir):
    """build_gaussian_pyramid with scale_factor 0.6 / 0.4 / 0.75 / 0.3 (sigma = 1 / scale_factor): the oracle takes
    the Gaussian weights from libm's exp, the reference from NumPy's; tests/golden/pyramid_scales.npz (made by
    importing the reference, make_golden_scales.py) pins these four: equal value for value."""
    z = np.load(golden_dir / "pyramid_scales.npz")
    for sf in (0.6, 0.4, 0.75, 0.3):
        pyr = oracle.build_gaussian_pyramid(z["image"], 3, sf)
        for l, a in enumerate(pyr):
            np.testing.assert_array_equal(a, z[f"sf{sf}_level{l}"])


MORE_SCALES = (0.35, 0.45, 0.55, 0.65, 0.8, 0.9, 0.25, 0.7, 1.0 / 3.0)


def test_pyramid_any_scale_factor_equals_the_reference(oracle, golden_dir):
    """nine further scale factors nobody tuned for (tests/golden/pyramid_scales_more.npz, made by importing the reference):
    with the Gaussian weights formed by NumPy as SciPy forms them (oracle.scipy_gaussian_weights) the oracle's pyramid is
    the reference's value for value -- no scale factor is "parity unpinned" any more.  Does libm's exp ever differ?  Counted
    here over these sigmas (the all-C fallback form), reported, not required."""
    z = np.load(golden_dir / "pyramid_scales_more.npz")
    differing = 0
    for sf in MORE_SCALES:
        levels = 3 if sf >= 0.3 else 2
        pyr = oracle.build_gaussian_pyramid(z["image"], levels, sf)
        alt = oracle.build_gaussian_pyramid_libm(z["image"], levels, sf)
        for l, a in enumerate(pyr):
            np.testing.assert_array_equal(a, z[f"sf{sf!r}_level{l}"])
            differing += int(np.count_nonzero(a != alt[l]))
    print(f"values where libm-exp weights change the pyramid: {differing}")


def test_windows_outside_3_to_11_match_the_reference(oracle, golden_dir):
    """window sizes the reference accepts beyond the tiled kernels' range (1x1; 13x13 and larger, where np.sum's
    pairwise order splits into blocks): the oracle reproduces the digests made by importing the reference
    (tests/golden/make_golden_windows.py) -- this pins the generic-window path's checker"""
    import hashlib
    import json

    ref = json.loads((golden_dir / "reference_windows.json").read_text())
    z = np.load(golden_dir / "patterns_320x240.npz")
    (y0, y1), (x0, x1) = ref["crop"]

    def digest(a):
        return hashlib.sha256((np.ascontiguousarray(a, np.float32) + np.float32(0.0)).tobytes()).hexdigest()

    for name, e in ref["patterns"].items():
        p = np.ascontiguousarray(z["frame_0"].astype(np.float32)[y0:y1, x0:x1])
        c = np.ascontiguousarray(z[f"frame_1__{name}"].astype(np.float32)[y0:y1, x0:x1])
        for win, d in e["single_scale"].items():
            u, v = oracle.lucas_kanade_single_scale(p, c, int(win))
            assert digest(u) == d["u_sha256"] and digest(v) == d["v_sha256"], (name, win)
            assert int(np.count_nonzero(u)) == d["nonzero_u"]
        for win, d in e["pyramidal"].items():
            u, v = oracle.lucas_kanade_pyramidal(p, c, d["levels"], int(win), d["iterations"])
            assert digest(u) == d["u_sha256"] and digest(v) == d["v_sha256"], (name, win)


FULLSIZE_KEYS = ["c1", "c2", "c3", "c4", "m1", "m2", "m3", "m4", "m5", "m6", "m7", "e1", "e2", "e3"]


def _fullsize_frames(c):
    from oflk_synth import synth_pair, synth_pair_smooth

    gen = synth_pair_smooth if c.get("smooth") else synth_pair
    return gen(c["shape"][0], c["shape"][1], c["pair_index"], c.get("dx", 3.0), c.get("dy", -1.5))


def _fullsize_cases(golden_dir):
    import json

    return json.loads((golden_dir / "reference_fullsize.json").read_text())


@pytest.mark.parametrize("key", FULLSIZE_KEYS)
def test_oracle_equals_the_reference_at_baseline_sizes(oracle, golden_dir, key):
    """BASELINE.json configs[1] (640x480 single-scale), configs[2] (1920x1080, 3 levels), one pair of configs[3]
    (3840x2160) and configs[4] in the reference's own fp32 (7680x4320, 7x7 single-scale) on the bench workload's synthetic
    frames, mid-size cases with other parameters (m*) and small motions whose levels exit early (e*): the oracle's flow and
    iteration counts equal the reference's own (tests/golden/reference_fullsize.json, made by importing the reference:
    minutes per pair there, seconds here)."""
    cases = _fullsize_cases(golden_dir)
    if key not in cases:
        pytest.skip(f"{key} not in reference_fullsize.json")
    c = cases[key]
    p, q = _fullsize_frames(c)
    if c["mode"] == "single_scale":
        u, v = oracle.lucas_kanade_single_scale(p, q, c["window_size"])
    else:
        u, v, _, runs = oracle.lucas_kanade_pyramidal_ex(p, q, c["levels"], c["window_size"], c["iterations"])
        if "iters_run" in c:
            assert list(runs) == c["iters_run"]
    assert digest(u) == c["u_sha256"] and digest(v) == c["v_sha256"]


@pytest.mark.parametrize("key", ["1080p", "4k", "odd"])
def test_oracle_stage_functions_equal_the_reference_at_full_size(oracle, golden_dir, key):
    """compute_gradients, build_gaussian_pyramid, warp_image, upsample_flow at 1920x1080, 3840x2160 and 1081x1923:
    digests of the reference's own output (tests/golden/reference_stages_fullsize.json, make_golden_stages_fullsize.py)."""
    import json

    from oflk_synth import synth_flow, synth_pair

    r = json.loads((golden_dir / "reference_stages_fullsize.json").read_text())[key]
    h, w = r["shape"]
    p, c = synth_pair(h, w, pair_index=0)
    ix, iy, it = oracle.compute_gradients(p, c)
    assert [digest(ix), digest(iy), digest(it)] == r["gradients"]
    pyr = oracle.build_gaussian_pyramid(c, 3)
    assert [list(a.shape) for a in pyr] == r["pyramid_shapes"] and [digest(a) for a in pyr] == r["pyramid"]
    fu, fv = synth_flow(h, w, seed=1)
    assert digest(oracle.warp_image(c, fu, fv)) == r["warp"]
    hc, wc = r["upsample_from"]
    cu, cv = synth_flow(hc, wc, seed=2)
    uu, uv = oracle.upsample_flow(cu, cv, (h, w))
    assert [digest(uu), digest(uv)] == r["upsample"]

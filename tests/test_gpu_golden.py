"""GPU tests against the reference's own outputs: the HIP path on the 13
verification patterns must reproduce the sha256 digests of the flow fields the
reference produced (tests/golden/reference_13patterns.json, made by importing the
reference), i.e. EPE vs the Python reference is exactly 0 (bar in BASELINE.json:
1e-4), and the verifier counterpart must reproduce verification_baseline.json."""
import hashlib
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PATTERNS = ["translate_small", "translate_medium", "translate_large", "translate_vertical", "translate_diagonal",
            "rotate_small", "rotate_medium", "rotate_large", "zoom_in", "zoom_out", "translate_rotate", "no_motion",
            "translate_extreme"]


def digest(a):
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


@pytest.fixture(scope="module")
def suite(golden_dir):
    return (np.load(golden_dir / "patterns_320x240.npz"),
            json.loads((golden_dir / "reference_13patterns.json").read_text()))


@pytest.mark.parametrize("name", PATTERNS)
def test_hip_flow_equals_reference_flow(suite, name):
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P

    z, ref = suite
    p, c = z["frame_0"].astype(np.float32), z[f"frame_1__{name}"].astype(np.float32)
    r = ref["patterns"][name]
    u, v = K.lucas_kanade_single_scale(p, c, 5)
    assert digest(u) == r["single_scale"]["u_sha256"] and digest(v) == r["single_scale"]["v_sha256"]
    u, v, log, runs = P.lucas_kanade_pyramidal_with_log(p, c, 3, 5, 3)
    assert list(runs) == r["pyramidal"]["iters_run"]
    for l, rows in enumerate(r["pyramidal"]["residual_log"]):
        # device means are an fp64 reduction; the reference's are an fp32 pairwise sum
        np.testing.assert_allclose(log[l, :len(rows)], np.array(rows, np.float32), rtol=2e-6, atol=1e-9)
    assert digest(u) == r["pyramidal"]["u_sha256"] and digest(v) == r["pyramidal"]["v_sha256"]


def test_epe_vs_reference_dense(golden_dir, suite):
    """the BASELINE.json parity metric, stated explicitly: mean EPE vs the reference flow <= 1e-4"""
    import lucas_kanade_pyramidal as P

    z, _ = suite
    d = np.load(golden_dir / "dense_translate_medium.npz")
    p, c = z["frame_0"].astype(np.float32), z["frame_1__translate_medium"].astype(np.float32)
    u, v = P.lucas_kanade_pyramidal(p, c, 3, 5, 3)
    epe = float(np.mean(np.sqrt((u.astype(np.float64) - d["pyr_u"]) ** 2 + (v.astype(np.float64) - d["pyr_v"]) ** 2)))
    assert epe <= 1e-4  # tolerance of the north star
    assert epe == 0.0   # what this build actually delivers


def test_verifier_on_gpu_reproduces_baseline(tmp_path, golden_dir, suite, monkeypatch):
    """the reference's CI recipe end to end on the HIP path: suite files -> verifier -> regression gate"""
    import generate_test_suite as G
    import optical_flow_verifier as V
    from conftest import PRODUCT

    z, ref = suite
    suite_dir = tmp_path / "python" / "test_suite"
    for name, params in G.TEST_PATTERNS.items():
        pdir = suite_dir / name
        pdir.mkdir(parents=True)
        z["frame_0"].tofile(pdir / "frame_00.bin")
        z[f"frame_1__{name}"].tofile(pdir / "frame_01.bin")
        (pdir / "metadata.json").write_text(json.dumps({"resolution": {"width": 320, "height": 240},
                                                        "motion_parameters": params.to_dict()}))
    G.write_suite_index(suite_dir, 320, 240)
    cfg = V.load_config(PRODUCT / "verification_config.yaml")
    results = []
    for name in V.load_test_suite_index(suite_dir)["patterns"]:
        results.append(V.verify_pattern(name, V.load_test_pattern(suite_dir / name), cfg, verbose=False))
    base = json.loads((golden_dir / "verification_baseline.json").read_text())["patterns"]
    for r in results:
        name = r["pattern_name"]
        for key in ("single_scale", "pyramidal"):
            assert r[key]["metrics"] == ref["patterns"][name][key]["metrics"], (name, key)
            assert r[key]["status"] == base[name][key]["status"], (name, key)
            for k, val in base[name][key]["metrics"].items():
                assert abs(r[key]["metrics"][k] - val) <= 2e-5
    assert V.compare_against_baseline(results, golden_dir / "verification_baseline.json", 10.0)


def test_rtl_frame_pairs_on_gpu(golden_dir):
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P

    r = np.load(golden_dir / "rtl_frames.npz")
    for tag in ("natural", "sinusoid"):
        p, c = r[f"{tag}__frame_00"].astype(np.float32), r[f"{tag}__frame_01"].astype(np.float32)
        us, vs = K.lucas_kanade_single_scale(p, c, 5)
        up, vp = P.lucas_kanade_pyramidal(p, c, 3, 5, 3)
        assert [digest(us), digest(vs), digest(up), digest(vp)] == list(r[f"{tag}__sha"])


@pytest.mark.parametrize("name", ["translate_medium", "rotate_small", "translate_extreme", "no_motion"])
@pytest.mark.parametrize("preset", ["shallow", "deep", "large_window"])
def test_hip_other_presets_equal_reference(suite, golden_dir, name, preset):
    """2-level, 4-level and 7x7 presets: HIP digests == the reference's digests"""
    import lucas_kanade_pyramidal as P

    z, _ = suite
    ref = json.loads((golden_dir / "reference_presets.json").read_text())[name][preset]
    p, c = z["frame_0"].astype(np.float32), z[f"frame_1__{name}"].astype(np.float32)
    u, v = P.lucas_kanade_pyramidal(p, c, ref["levels"], ref["window_size"], ref["iterations"])
    assert digest(u) == ref["u_sha256"] and digest(v) == ref["v_sha256"]


def test_demo_clis_on_rtl_frames(tmp_path, golden_dir, monkeypatch, capsys):
    """the two demo CLIs (reference lucas_kanade_reference.py / lucas_kanade_pyramidal.py main) on
    the frame pair the reference commits for its RTL testbench: raw float32 dumps equal the
    reference's flow (digests), region means equal the recorded answers"""
    import sys

    import lucas_kanade_pyramidal as P
    import lucas_kanade_reference as C

    r = np.load(golden_dir / "rtl_frames.npz")
    fdir = tmp_path / "frames"
    fdir.mkdir()
    r["natural__frame_00"].tofile(fdir / "frame_00.bin")
    r["natural__frame_01"].tofile(fdir / "frame_01.bin")
    out = tmp_path / "out"
    monkeypatch.setattr(sys, "argv", ["lucas_kanade_reference.py", "--frame-dir", str(fdir), "--output-dir", str(out)])
    monkeypatch.setattr(C, "visualize_flow", lambda *a, **k: None)
    C.main()
    u = np.fromfile(out / "flow_u.bin", np.float32).reshape(240, 320)
    v = np.fromfile(out / "flow_v.bin", np.float32).reshape(240, 320)
    sha = list(r["natural__sha"])
    assert [digest(u), digest(v)] == sha[:2]
    assert np.mean(u[105:135, 55:85]) == r["natural__answers"][0]
    monkeypatch.setattr(sys, "argv", ["lucas_kanade_pyramidal.py", "--frame-dir", str(fdir), "--output-dir", str(out)])
    P.main()
    up = np.fromfile(out / "flow_u_pyramidal.bin", np.float32).reshape(240, 320)
    vp = np.fromfile(out / "flow_v_pyramidal.bin", np.float32).reshape(240, 320)
    assert [digest(up), digest(vp)] == sha[2:]
    assert "Mean flow in test region" in capsys.readouterr().out


def test_uint8_ingestion_equals_float_path(suite):
    """raw 8-bit frames (the reference's .bin format) through oflk_*_u8: same flow as converting on the host"""
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P

    z, ref = suite
    p8, c8 = z["frame_0"], z["frame_1__translate_rotate"]
    r = ref["patterns"]["translate_rotate"]
    u, v = K.lucas_kanade_single_scale(p8, c8, 5)
    assert digest(u) == r["single_scale"]["u_sha256"] and digest(v) == r["single_scale"]["v_sha256"]
    u, v = P.lucas_kanade_pyramidal(p8, c8, 3, 5, 3)
    assert digest(u) == r["pyramidal"]["u_sha256"] and digest(v) == r["pyramidal"]["v_sha256"]
    # odd sizes / unaligned tails of the conversion kernel
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    b = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    u8 = K.lucas_kanade_single_scale(a, b, 5)
    uf = K.lucas_kanade_single_scale(a.astype(np.float32), b.astype(np.float32), 5)
    assert np.array_equal(u8[0], uf[0]) and np.array_equal(u8[1], uf[1])


@pytest.mark.parametrize("key", ["c1", "c2", "c3", "c4", "m1", "m2", "m3", "m4", "m5", "m6", "m7", "e1", "e2", "e3"])
def test_hip_equals_the_reference_at_baseline_sizes(golden_dir, key):
    """BASELINE.json configs[1], [2], one pair of [3] and [4] (its exact fp32 form) on the bench workload's frames, mid-size
    cases with other parameters and odd shapes (m*), small motions whose levels exit early (e*): HIP digests and iteration
    counts == those of the reference's own run (tests/golden/reference_fullsize.json); float32 and, where the frames are
    8-bit, uint8 arrays."""
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P
    from oflk_synth import synth_pair, synth_pair_smooth

    cases = json.loads((golden_dir / "reference_fullsize.json").read_text())
    if key not in cases:
        pytest.skip(f"{key} not in reference_fullsize.json")
    c = cases[key]
    gen = synth_pair_smooth if c.get("smooth") else synth_pair
    p, q = gen(c["shape"][0], c["shape"][1], c["pair_index"], c.get("dx", 3.0), c.get("dy", -1.5))
    forms = [(p, q)] + ([] if c.get("smooth") else [(p.astype(np.uint8), q.astype(np.uint8))])
    for frames in forms:
        if c["mode"] == "single_scale":
            u, v = K.lucas_kanade_single_scale(frames[0], frames[1], c["window_size"])
        else:
            u, v, _, runs = P.lucas_kanade_pyramidal_with_log(frames[0], frames[1], c["levels"], c["window_size"], c["iterations"])
            if "iters_run" in c:
                assert list(runs[:c["levels"]]) == c["iters_run"]
        assert digest(u) == c["u_sha256"] and digest(v) == c["v_sha256"]


@pytest.mark.parametrize("key", ["1080p", "4k", "odd"])
def test_hip_stage_functions_equal_the_reference_at_full_size(golden_dir, key):
    """compute_gradients, build_gaussian_pyramid, warp_image, upsample_flow at 1920x1080, 3840x2160 and 1081x1923:
    digests of the reference's own output (tests/golden/reference_stages_fullsize.json, make_golden_stages_fullsize.py)."""
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P
    from oflk_synth import synth_flow, synth_pair

    r = json.loads((golden_dir / "reference_stages_fullsize.json").read_text())[key]
    h, w = r["shape"]
    p, c = synth_pair(h, w, pair_index=0)
    ix, iy, it = K.compute_gradients(p, c)
    assert [digest(ix), digest(iy), digest(it)] == r["gradients"]
    pyr = P.build_gaussian_pyramid(c, 3)
    assert [list(a.shape) for a in pyr] == r["pyramid_shapes"] and [digest(a) for a in pyr] == r["pyramid"]
    fu, fv = synth_flow(h, w, seed=1)
    assert digest(P.warp_image(c, fu, fv)) == r["warp"]
    hc, wc = r["upsample_from"]
    cu, cv = synth_flow(hc, wc, seed=2)
    uu, uv = P.upsample_flow(cu, cv, (h, w))
    assert [digest(uu), digest(uv)] == r["upsample"]

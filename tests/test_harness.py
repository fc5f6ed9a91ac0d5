"""CPU tests of the host-side harness counterparts (pattern generator, metrics,
verifier).  BASELINE config 1 -- `translate_small`, single-scale, driven through
the verifier on CPU -- runs here with the ORACLE injected in place of the HIP
functions: that is test plumbing (the shipped verifier imports the HIP drop-ins
and has no CPU path)."""
import hashlib
import json

import numpy as np
import pytest


@pytest.fixture(scope="module")
def V():
    import optical_flow_verifier

    return optical_flow_verifier


@pytest.fixture(scope="module")
def cfg(V):
    from conftest import PRODUCT

    return V.load_config(PRODUCT / "verification_config.yaml")


def test_generator_reproduces_committed_patterns(golden_dir):
    pytest.importorskip("PIL")
    import generate_test_suite as G

    z = np.load(golden_dir / "patterns_320x240.npz")
    base = G.load_base_texture(320, 240)
    assert np.array_equal(base, z["frame_0"])
    assert list(G.TEST_PATTERNS) == ["translate_small", "translate_medium", "translate_large", "translate_vertical",
                                     "translate_diagonal", "rotate_small", "rotate_medium", "rotate_large", "zoom_in",
                                     "zoom_out", "translate_rotate", "no_motion", "translate_extreme"]
    for name, params in G.TEST_PATTERNS.items():
        _, f1 = G.generate_test_pattern(params, base=base)
        assert np.array_equal(f1, z[f"frame_1__{name}"]), name
    # digests recorded in SURVEY.md Appendix B (reference-run verified there)
    assert hashlib.sha256(base.tobytes()).hexdigest()[:12] == "29964567a77c"
    assert hashlib.sha256(z["frame_1__translate_extreme"].tobytes()).hexdigest()[:12] == "152cb022c47e"


def test_suite_files_round_trip(tmp_path, V, golden_dir):
    pytest.importorskip("PIL")
    import generate_test_suite as G

    z = np.load(golden_dir / "patterns_320x240.npz")
    G.generate_full_suite(320, 240, tmp_path, base=z["frame_0"], save_png=False)
    index = V.load_test_suite_index(tmp_path)
    assert index["num_patterns"] == 13 and list(index["patterns"]) == list(G.TEST_PATTERNS)
    d = V.load_test_pattern(tmp_path / "translate_small")
    assert d["frame_prev"].dtype == np.float32 and d["frame_prev"].shape == (240, 320)
    assert np.array_equal(d["frame_curr"], z["frame_1__translate_small"].astype(np.float32))
    assert d["metadata"]["motion_parameters"]["dx"] == 0.5
    mem = (tmp_path / "translate_small" / "frame_00.mem").read_text().split()
    assert len(mem) == 320 * 240 and int(mem[0], 16) == int(z["frame_0"][0, 0])


def test_masks_and_classification(V, cfg):
    assert int(V.get_test_region_mask((240, 320), "translate_medium", 80).sum()) == 66000
    assert int(V.get_test_region_mask((240, 320), "rotate_small", 80).sum()) == 6400
    assert int(V.get_test_region_mask((240, 320), "translate_rotate", 80).sum()) == 6400
    assert V.classify_result(0.4, 0.5, "translate_small", cfg) == "Pass"
    assert V.classify_result(0.4, 0.51, "translate_small", cfg) == "Warning"
    assert V.classify_result(2.1, 0.0, "translate_small", cfg) == "Fail"
    assert V.classify_result(2.9, 0.0, "zoom_in", cfg) == "Warning"
    assert V.get_thresholds_for_pattern("translate_rotate", cfg) == (2.0, 5.0)


def test_compare_metrics_rules(V):
    base = {"mae_u": 1.0, "mae_v": 0.0, "epe": 2.0}
    ok = V.compare_metrics({"mae_u": 1.09, "mae_v": 0.0, "epe": 1.85}, base, 10.0)
    assert ok["passed"] and set(ok["differences"]) == {"mae_u", "epe"}
    bad = V.compare_metrics({"mae_u": 1.11, "mae_v": 1e-3, "epe": 2.0}, base, 10.0)
    assert not bad["passed"] and len(bad["flags"]) == 2  # +11 % and "baseline was 0"


def test_config1_translate_small_through_the_verifier(V, cfg, oracle, golden_dir, monkeypatch, capsys):
    """BASELINE.json configs[0]: plumbing, no GPU -- oracle injected as the flow functions"""
    monkeypatch.setattr(V, "lucas_kanade_single_scale", oracle.lucas_kanade_single_scale)
    monkeypatch.setattr(V, "lucas_kanade_pyramidal",
                        lambda p, c, num_levels, window_size, num_iterations:
                        oracle.lucas_kanade_pyramidal(p, c, num_levels, window_size, num_iterations))
    z = np.load(golden_dir / "patterns_320x240.npz")
    ref = json.loads((golden_dir / "reference_13patterns.json").read_text())["patterns"]["translate_small"]
    data = {"frame_prev": z["frame_0"].astype(np.float32), "frame_curr": z["frame_1__translate_small"].astype(np.float32),
            "metadata": {"motion_parameters": ref["motion"], "resolution": {"width": 320, "height": 240}}}
    res = V.verify_pattern("translate_small", data, cfg, verbose=True)
    assert res["num_test_pixels"] == 66000
    assert res["single_scale"]["metrics"] == ref["single_scale"]["metrics"]
    assert res["pyramidal"]["metrics"] == ref["pyramidal"]["metrics"]
    base = json.loads((golden_dir / "verification_baseline.json").read_text())
    assert res["single_scale"]["status"] == base["patterns"]["translate_small"]["single_scale"]["status"] == "Pass"
    assert res["pyramidal"]["status"] == base["patterns"]["translate_small"]["pyramidal"]["status"] == "Warning"
    assert abs(res["single_scale"]["metrics"]["epe"] - 0.39136627316474915) < 2e-5  # verification_baseline.json:17
    # regression gate against the reference's committed baseline passes
    assert V.compare_against_baseline([res], golden_dir / "verification_baseline.json", 10.0)
    table = V.generate_markdown_table([res])
    assert "| translate_small      | ( 0.5,  0.5) | 0.265 | 0.245 | 0.466 | 0.391 | 13.60° | Pass |" in table
    assert "Testing: translate_small" in capsys.readouterr().out


def test_flow_metrics_definitions():
    import flow_metrics as M

    u = np.array([[1.0, 3.0], [2.0, 2.0]], np.float32)
    v = np.array([[0.0, 0.0], [1.0, -1.0]], np.float32)
    m = M.compute_all_metrics(u, v, 2.0, 0.0)
    assert m["mae_u"] == 0.5 and m["mae_v"] == 0.5
    assert abs(m["epe"] - 1.0) < 1e-7 and abs(m["rmse"] - 1.0) < 1e-7
    z = np.zeros((4, 4), np.float32)
    assert M.angular_error(z, z, 0.0, 0.0) == 0.0
    mask = np.zeros((2, 2), bool)
    mask[0, 0] = True
    assert M.mean_absolute_error(u, v, 2.0, 0.0, mask) == (1.0, 0.0)


def test_flow_text_dump_format(tmp_path):
    """`x y u v` dump shared with scripts/visualize_flow.py and the RTL testbench (reference
    lucas_kanade_reference.py:78-103)"""
    import lucas_kanade_reference as C

    u = np.arange(6, dtype=np.float32).reshape(2, 3) / 4
    v = -u
    C.export_flow_field_txt(u, v, tmp_path / "f.txt", 3, 2, {"x_min": 0, "x_max": 1, "y_min": 0, "y_max": 1})
    lines = (tmp_path / "f.txt").read_text().splitlines()
    assert lines[:4] == ["# Optical flow field data (Python reference)", "# Format: x y u v", "# Image size: 3x2",
                         "# Test region: x[0:1], y[0:1]"]
    assert lines[4] == "0 0 0.000000 -0.000000" and lines[-1] == "2 1 1.250000 -1.250000" and len(lines) == 10


def test_mask_rectangle_of_the_verifier_regions():
    """flow_metrics.mask_rectangle recovers the slice bounds of get_test_region_mask's masks"""
    import flow_metrics as M
    import optical_flow_verifier as V

    assert M.mask_rectangle(V.get_test_region_mask((240, 320), "translate_small", 80)) == (10, 230, 10, 310)
    assert M.mask_rectangle(V.get_test_region_mask((240, 320), "rotate_small", 80)) == (80, 160, 120, 200)
    assert M.mask_rectangle(np.zeros((4, 4), bool)) == (0, 0, 0, 0)
    bad = np.ones((6, 6), bool)
    bad[2, 3] = False
    with pytest.raises(ValueError):
        M.mask_rectangle(bad)


def test_visualize_flow_parses_dumps_and_plots(tmp_path):
    """counterpart of the reference's scripts/visualize_flow.py: the `x y u v` dump written by
    lucas_kanade_reference.export_flow_field_txt round-trips through parse_flow_field, the test-region
    statistics use the reference's inclusive slice and skip zero vectors, and the 4-panel PNG is written"""
    import lucas_kanade_reference as LR
    import visualize_flow as VF

    H, W = 24, 32
    rng = np.random.default_rng(0)
    u = rng.normal(2.0, 0.1, (H, W)).astype(np.float32)
    v = rng.normal(0.0, 0.1, (H, W)).astype(np.float32)
    u[:2] = 0
    v[:2] = 0                                   # border rows without flow
    region = {"x_min": 5, "x_max": 20, "y_min": 1, "y_max": 10}
    dump = tmp_path / "flow_field.txt"
    LR.export_flow_field_txt(u, v, dump, W, H, region)
    x, y, uu, vv, meta = VF.parse_flow_field(str(dump))
    assert meta == {"width": W, "height": H, "test_x_min": 5, "test_x_max": 20, "test_y_min": 1, "test_y_max": 10}
    assert len(x) == H * W and np.allclose(uu.reshape(H, W), u, atol=1e-6) and np.allclose(vv.reshape(H, W), v, atol=1e-6)
    uf, vf, mf = VF.flow_grids(x, y, uu, vv, H, W)
    st = VF.region_statistics(uu, vv, uf, vf, mf, meta)
    sl = (slice(1, 11), slice(5, 21))           # inclusive bounds, like the reference (:113-118)
    keep = np.hypot(uf, vf)[sl] > 0
    assert st["num_vectors"] == int(keep.sum()) == 9 * 16   # row 1 carries no flow
    assert abs(st["mean_u"] - uf[sl][keep].mean()) < 1e-12 and abs(st["std_v"] - vf[sl][keep].std()) < 1e-12
    # no region in the header: statistics over every vector
    LR.export_flow_field_txt(u, v, dump, W, H, None)
    x, y, uu, vv, meta = VF.parse_flow_field(str(dump))
    assert "test_x_min" not in meta
    st = VF.region_statistics(uu, vv, *VF.flow_grids(x, y, uu, vv, H, W), meta)
    assert st["num_vectors"] == H * W
    try:
        import matplotlib  # noqa: F401
        from PIL import Image
    except ImportError:
        return
    frame = tmp_path / "frame_00.png"
    Image.fromarray(rng.integers(0, 256, (H, W), dtype=np.uint8)).save(frame)
    LR.export_flow_field_txt(u, v, dump, W, H, region)
    out = tmp_path / "results" / "viz.png"
    VF.create_diagnostic_plot(str(frame), str(dump), str(out), 2.0, 0.0, stride=7)
    assert out.exists() and out.stat().st_size > 10_000


def test_markdown_report_equals_the_reference_held_report(golden_dir):
    """generate_markdown_table on the metrics the reference published (verification_baseline.json) reproduces the report
    the reference holds next to them (python/verification_results.md, copied to tests/golden/ as data): same rows,
    same number formats, same status words, same legend"""
    import json

    import optical_flow_verifier as V

    base = json.loads((golden_dir / "verification_baseline.json").read_text())
    results = list(base["patterns"].values())
    ours = V.generate_markdown_table(results).strip().splitlines()
    theirs = (golden_dir / "verification_results.md").read_text().strip().splitlines()
    rows = lambda ls: [l.rstrip() for l in ls if l.strip()]
    assert rows(ours) == rows(theirs)

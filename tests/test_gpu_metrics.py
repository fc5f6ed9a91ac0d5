"""GPU tests of the device-side masked metrics: oflk_flow_metrics /
oflk_plan_metrics against the host flow_metrics module (the reference's arithmetic,
python/flow_metrics.py:14-201) and against the reference's own verification_baseline.json.

Tolerance: the device adds up in fp64 and uses an fp64 arccos, the reference takes fp32 pairwise
means and NumPy's fp32 arccos; 1e-5 relative (+1e-6 absolute) covers that, the baseline file is
held to the 2e-5 the verifier test uses."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("mae_u", "mae_v", "rmse", "epe", "aae")


def _close(dev, host):
    for k in KEYS:
        assert dev[k] == pytest.approx(host[k], rel=1e-5, abs=1e-6), (k, dev[k], host[k])


@pytest.mark.parametrize("shape,region,truth", [
    ((240, 320), (10, -10, 10, -10), (2.0, 0.0)),        # translation patterns: frame minus a 10 px border
    ((240, 320), (70, 170, 110, 210), (0.0, 0.0)),       # rotation / zoom: centred crop
    ((67, 91), (0, 67, 0, 91), (-1.5, 0.75)),
    ((33, 40), (5, 6, 7, 39), (15.0, -3.0)),             # a single row
])
def test_device_metrics_match_host_metrics(shape, region, truth):
    import flow_metrics as M

    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    u = (truth[0] + rng.normal(0, 1.5, shape)).astype(np.float32)
    v = (truth[1] + rng.normal(0, 0.7, shape)).astype(np.float32)
    mask = np.zeros(shape, bool)
    mask[region[0]:region[1], region[2]:region[3]] = True
    _close(M.compute_all_metrics_gpu(u, v, truth[0], truth[1], mask), M.compute_all_metrics(u, v, truth[0], truth[1], mask))
    _close(M.compute_all_metrics_gpu(u, v, truth[0], truth[1]), M.compute_all_metrics(u, v, truth[0], truth[1]))


def test_nothing_moves_nothing_predicted():
    """flow_metrics.py:143-146: zero truth and zero prediction define the angular error as 0"""
    import flow_metrics as M

    z = np.zeros((48, 64), np.float32)
    m = M.compute_all_metrics_gpu(z, z, 0.0, 0.0)
    assert m == {k: 0.0 for k in KEYS}
    z[10, 10] = 1e-3   # one pixel predicted to move: the general formula applies again
    _close(M.compute_all_metrics_gpu(z, z, 0.0, 0.0), M.compute_all_metrics(z, z, 0.0, 0.0))


def test_non_rectangular_mask_is_refused():
    import flow_metrics as M

    mask = np.zeros((20, 20), bool)
    mask[2:8, 2:8] = True
    mask[4, 4] = False
    z = np.zeros((20, 20), np.float32)
    with pytest.raises(ValueError):
        M.compute_all_metrics_gpu(z, z, 1.0, 0.0, mask)


def test_device_metrics_reproduce_the_reference_baseline(golden_dir):
    """13 patterns x {single-scale, pyramidal}: flows from the HIP path, metrics reduced on the device,
    compared with the numbers the reference committed (python/verification_baseline.json)."""
    import flow_metrics as M
    import generate_test_suite as G
    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P
    import optical_flow_verifier as V

    z = np.load(golden_dir / "patterns_320x240.npz")
    base = json.loads((golden_dir / "verification_baseline.json").read_text())["patterns"]
    p = z["frame_0"].astype(np.float32)
    for name, params in G.TEST_PATTERNS.items():
        c = z[f"frame_1__{name}"].astype(np.float32)
        mask = V.get_test_region_mask(p.shape, name, 80)   # verification_config.yaml test_region.center_crop
        motion = params.to_dict()
        for key, (u, v) in (("single_scale", K.lucas_kanade_single_scale(p, c, 5)),
                            ("pyramidal", P.lucas_kanade_pyramidal(p, c, 3, 5, 3))):
            m = M.compute_all_metrics_gpu(u, v, motion["dx"], motion["dy"], mask)
            for k, val in base[name][key]["metrics"].items():
                assert abs(m[k] - val) <= 2e-5, (name, key, k, m[k], val)


class _DevBuf:
    """device memory through the HIP runtime liboflk is already linked to (keeps this file torch-free)"""

    def __init__(self, arr=None, nbytes=0):
        import ctypes

        import _oflk

        _oflk.lib()
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.ptr = ctypes.c_void_p()
        self.nbytes = arr.nbytes if arr is not None else nbytes
        assert self.hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(self.nbytes)) == 0
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            assert self.hip.hipMemcpy(self.ptr, arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes), 1) == 0

    def to_host(self, shape, dtype=np.float32):
        import ctypes

        out = np.empty(shape, dtype)
        assert self.hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.ptr, ctypes.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def test_plan_metrics_on_device_resident_batch():
    """oflk_plan_metrics: flows never leave the GPU; one row of five numbers per pair comes back"""
    import _oflk
    import flow_metrics as M
    from oflk_synth import synth_pair

    B, H, W = 3, 120, 160
    pairs = [synth_pair(H, W, i, dx=1.0 + i, dy=-0.5 * i) for i in range(B)]
    prev = _DevBuf(np.stack([a for a, _ in pairs]))
    curr = _DevBuf(np.stack([b for _, b in pairs]))
    u, v = _DevBuf(nbytes=prev.nbytes), _DevBuf(nbytes=prev.nbytes)
    plan = _oflk.Plan(0, B, H, W, 3, 5, 3)
    plan.pyramidal(prev.ptr.value, curr.ptr.value, u.ptr.value, v.ptr.value, 0)
    ut = [1.0 + i for i in range(B)]
    vt = [-0.5 * i for i in range(B)]
    got = plan.metrics(u.ptr.value, v.ptr.value, ut, vt, (10, -10, 10, -10), 0)
    plan.close()
    hu, hv = u.to_host((B, H, W)), v.to_host((B, H, W))
    for buf in (prev, curr, u, v):
        buf.free()
    mask = np.zeros((H, W), bool)
    mask[10:-10, 10:-10] = True
    for b in range(B):
        _close(dict(zip(KEYS, got[b])), M.compute_all_metrics(hu[b], hv[b], ut[b], vt[b], mask))
    # and the plan path gives the flows the host entry point gives
    import lucas_kanade_pyramidal as P

    for b in range(B):
        eu, ev = P.lucas_kanade_pyramidal(pairs[b][0], pairs[b][1], 3, 5, 3)
        assert np.array_equal(hu[b], eu) and np.array_equal(hv[b], ev)


def test_verifier_with_device_metrics_keeps_every_status(golden_dir):
    """optical_flow_verifier.verify_pattern(device_metrics=True): same Pass/Warning/Fail classification and
    metrics within 2e-5 of the reference's baseline on all 13 patterns"""
    import generate_test_suite as G
    import optical_flow_verifier as V
    from conftest import PRODUCT

    z = np.load(golden_dir / "patterns_320x240.npz")
    base = json.loads((golden_dir / "verification_baseline.json").read_text())["patterns"]
    cfg = V.load_config(PRODUCT / "verification_config.yaml")
    p = z["frame_0"].astype(np.float32)
    for name, params in G.TEST_PATTERNS.items():
        data = {"frame_prev": p, "frame_curr": z[f"frame_1__{name}"].astype(np.float32),
                "metadata": {"motion_parameters": params.to_dict()}}
        r = V.verify_pattern(name, data, cfg, verbose=False, device_metrics=True)
        for key in ("single_scale", "pyramidal"):
            assert r[key]["status"] == base[name][key]["status"], (name, key)
            for k, val in base[name][key]["metrics"].items():
                assert abs(r[key]["metrics"][k] - val) <= 2e-5, (name, key, k)

"""RTL-bit-accurate integer mode on the GPU: `oflk_rtl_flow_u8` and the
testbench bookkeeping of rtl_golden_model.py against the oracle (oracle/rtl_model.py, held equal to the
cycle-by-cycle execution of the RTL by tests/test_rtl_model.py).  Integer work: the bar is bit-exact.
PARITY UNPINNED (no simulator output of the RTL as committed exists): see include/oflk.h."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_states(f0, f1):
    import rtl_model as M

    gx, gy, gt, _ = M.gradient_stream(f0, f1)
    return M.flow_states(gx, gy, gt, f0.shape[1])


@pytest.mark.parametrize("shape", [(5, 5), (6, 9), (12, 16), (31, 17), (64, 48), (240, 320), (37, 1024), (512, 23), (512, 1024)])
def test_per_element_flow_is_bit_exact(shape):
    import rtl_golden_model as G

    H, W = shape
    rng = np.random.default_rng(H + 7 * W)
    yy, xx = np.mgrid[0:H, 0:W]
    tex = (128 + 100 * np.sin(xx / 3.0) * np.cos(yy / 4.0)).astype(np.int64)
    for f0, f1 in ((rng.integers(0, 256, (H, W)), rng.integers(0, 256, (H, W))), (tex, np.roll(tex, (1, 2), (0, 1)))):
        f0, f1 = f0.astype(np.uint8), f1.astype(np.uint8)
        st = G.rtl_flow_states(f0, f1)
        valid, x, y, u, v = _oracle_states(f0, f1)
        assert np.array_equal(st["valid"], valid) and np.array_equal(st["x"], x) and np.array_equal(st["y"], y)
        assert np.array_equal(st["u"].astype(np.int64), u) and np.array_equal(st["v"].astype(np.int64), v), shape


def test_batch_equals_per_pair_and_device_entry():
    import torch

    import _oflk
    import rtl_golden_model as G

    rng = np.random.default_rng(3)
    B, H, W = 5, 50, 70
    p = rng.integers(0, 256, (B, H, W)).astype(np.uint8)
    c = rng.integers(0, 256, (B, H, W)).astype(np.uint8)
    st = G.rtl_flow_states(p, c)
    for b in range(B):
        one = G.rtl_flow_states(p[b], c[b])
        assert np.array_equal(st["u"][b], one["u"]) and np.array_equal(st["v"][b], one["v"])
    dev = torch.device("cuda", 0)
    tp, tc = torch.from_numpy(p).to(dev), torch.from_numpy(c).to(dev)
    M = int(_oflk.lib().oflk_rtl_stream_length(H, W))
    assert M == (H - 4) * (W - 4)
    du = torch.zeros((B, M), dtype=torch.int16, device=dev)
    dv = torch.zeros_like(du)
    _oflk.check(_oflk.lib().oflk_rtl_flow_u8_device(tp.data_ptr(), tc.data_ptr(), B, H, W, du.data_ptr(), dv.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(du.cpu().numpy(), st["u"]) and np.array_equal(dv.cpu().numpy(), st["v"])


@pytest.mark.parametrize("which", ["sinusoid", "natural"])
def test_testbench_vectors_equal_the_cycle_simulation(golden_dir, which, tmp_path):
    """the two 320x240 pairs of tests/golden/rtl_frames.npz (the testbench's own frames and the generator's
    sinusoid): every sampled vector, the summary and the exported flow_field.txt"""
    import rtl_cycle_sim as S
    import rtl_golden_model as G
    import visualize_flow as V

    z = np.load(golden_dir / "rtl_frames.npz")
    f0, f1 = z[f"{which}__frame_00"], z[f"{which}__frame_01"]
    sim = np.array(S.simulate(f0.reshape(-1).astype(np.int64), f1.reshape(-1).astype(np.int64), 320, 240), np.int64)
    vec = G.testbench_vectors(f0, f1)
    assert np.array_equal(vec, sim)
    s = G.testbench_summary(vec)
    assert s["valid_flow_count"] == 73287 and s["first_vector_position"] == (3, 2)
    # the testbench's own statistics loop, restated: positions lag by one sample
    px, py, n, su = sim[0, 0], sim[0, 1], 0, 0.0
    for x, y, u, _ in sim:
        if 55 <= px <= 85 and 105 <= py <= 135:
            n += 1
            su += u / 128.0
        px, py = x, y
    assert s["test_region_count"] == n and abs(s["mean_u"] - su / n) < 1e-12
    out = tmp_path / "flow_field.txt"
    G.write_flow_field(out, vec, 320, 240)
    x, y, u, v, meta = V.parse_flow_field(str(out))
    assert meta == {"width": 320, "height": 240, "test_x_min": 55, "test_x_max": 85, "test_y_min": 105, "test_y_max": 135}
    assert len(x) == len(vec) and np.allclose(u, vec[:, 2] / 128.0, atol=1e-6) and np.array_equal(x[1:].astype(int), vec[:-1, 0])


def test_cli_on_mem_frames(golden_dir, tmp_path, capsys):
    import rtl_golden_model as G

    z = np.load(golden_dir / "rtl_frames.npz")
    for i in (0, 1):
        (tmp_path / f"frame_0{i}.mem").write_text("\n".join("%02x" % p for p in z[f"sinusoid__frame_0{i}"].reshape(-1)) + "\n")
    assert G.main([str(tmp_path / "frame_00.mem"), str(tmp_path / "frame_01.mem"), "--output", str(tmp_path / "ff.txt")]) == 0
    out = capsys.readouterr().out
    assert "Total valid flow vectors: 73287" in out and "Results Summary" in out
    assert (tmp_path / "ff.txt").read_text().startswith("# Optical flow field data\n# Format: x y u v\n# Image size: 320x240\n")


def test_compare_with_a_simulator_export(golden_dir, tmp_path, capsys):
    """the one command that pins the model once somebody has a simulator: --compare flow_field.txt"""
    import rtl_golden_model as G

    z = np.load(golden_dir / "rtl_frames.npz")
    for i in (0, 1):
        z[f"sinusoid__frame_0{i}"].tofile(tmp_path / f"frame_0{i}.bin")
    args = [str(tmp_path / "frame_00.bin"), str(tmp_path / "frame_01.bin")]
    assert G.main(args + ["--output", str(tmp_path / "sim.txt")]) == 0
    assert G.main(args + ["--compare", str(tmp_path / "sim.txt")]) == 0
    assert "MODEL PINNED" in capsys.readouterr().out
    lines = (tmp_path / "sim.txt").read_text().splitlines()
    lines[1000] = lines[1000].rsplit(" ", 1)[0] + " 0.507812"     # one vector off by a few LSBs
    (tmp_path / "sim2.txt").write_text("\n".join(lines[:-2]) + "\n")   # and two vectors short, like the README's count
    assert G.main(args + ["--compare", str(tmp_path / "sim2.txt")]) == 1
    out = capsys.readouterr().out
    assert "MODEL DIFFERS" in out and "differing lines: 3" in out


def test_errors_are_loud():
    import _oflk
    import rtl_golden_model as G

    a = np.zeros((8, 8), np.uint8)
    with pytest.raises(ValueError):
        G.rtl_flow_states(a.astype(np.float32), a.astype(np.float32))
    with pytest.raises(ValueError):
        G.rtl_flow_states(np.zeros((4, 8), np.uint8), np.zeros((4, 8), np.uint8))       # fewer rows than the line buffers hold
    with pytest.raises(_oflk.OflkError):
        G.rtl_flow_states(np.zeros((8, 1025), np.uint8), np.zeros((8, 1025), np.uint8))  # flow_x is 10 bits wide
    u = np.zeros(16, np.int16)
    assert _oflk.lib().oflk_rtl_flow_u8(None, None, 1, 8, 8, u.ctypes.data_as(ctypes.c_void_p), u.ctypes.data_as(ctypes.c_void_p)) != 0

#!/usr/bin/env python3
"""Dense flows of THE REFERENCE (imported here, build container only) for tolerance tests: a within-tolerance arithmetic
mode cannot be graded by digests, it needs the reference's values.
  * the 13 verification patterns (tests/golden/patterns_320x240.npz), 3 levels / 5x5 / 3 iterations
  * pair 0 of the bench workload at 1920x1080 (optical-flow-fpga_amd/python/oflk_synth.py), same parameters (~4 min)
Each field is stored as float32 u, v plus the iteration counts per level read from the lines the reference prints.
Also checks every field against the digests already committed (reference_13patterns.json, reference_fullsize.json c2).
Output: tests/golden/dense_reference_flows.npz (compressed, ~20 MB).   Usage: python3 tests/golden/make_golden_dense.py"""
import contextlib
import hashlib
import importlib.util
import io
import json
import re
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/python")
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)

R_pyr.visualize_pyramid_level = lambda *a, **k: None


def digest(a):
    return hashlib.sha256((np.ascontiguousarray(a, np.float32) + np.float32(0.0)).tobytes()).hexdigest()


def run(p, c):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        u, v = R_pyr.lucas_kanade_pyramidal(p, c, num_levels=3, window_size=5, num_iterations=3)
    iters = [0, 0, 0]
    level = -1
    for line in buf.getvalue().splitlines():
        m = re.search(r"Processing pyramid level (\d+)/", line)
        if m:
            level = int(m.group(1))
        if "Iteration" in line and level >= 0:
            iters[level] = max(iters[level], int(re.search(r"Iteration (\d+)", line).group(1)))
    return u.astype(np.float32), v.astype(np.float32), np.array(iters, np.int32), buf.getvalue()


def main():
    out = {}
    z = np.load(HERE / "patterns_320x240.npz")
    ref = json.loads((HERE / "reference_13patterns.json").read_text())["patterns"]
    f0 = z["frame_0"].astype(np.float32)
    for k in z.files:
        if not k.startswith("frame_1__"):
            continue
        n = k[len("frame_1__"):]
        u, v, it, _ = run(f0, z[k].astype(np.float32))
        assert digest(u) == ref[n]["pyramidal"]["u_sha256"] and digest(v) == ref[n]["pyramidal"]["v_sha256"], n
        assert list(it) == ref[n]["pyramidal"]["iters_run"], (n, it, ref[n]["pyramidal"]["iters_run"])
        out[f"{n}__u"], out[f"{n}__v"], out[f"{n}__iters"] = u, v, it
        print(n, it, flush=True)
    spec = importlib.util.spec_from_file_location("oflk_synth", HERE.parents[1] / "optical-flow-fpga_amd" / "python" / "oflk_synth.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    p, c = m.synth_pair(1080, 1920, 0)
    u, v, it, _ = run(p, c)
    full = json.loads((HERE / "reference_fullsize.json").read_text())["c2"]
    assert digest(u) == full["u_sha256"] and digest(v) == full["v_sha256"], "1080p pair differs from the committed c2 digests"
    out["bench_1080p_pair0__u"], out["bench_1080p_pair0__v"], out["bench_1080p_pair0__iters"] = u, v, it
    print("bench_1080p_pair0", it, flush=True)
    np.savez_compressed(HERE / "dense_reference_flows.npz", **out)


if __name__ == "__main__":
    main()

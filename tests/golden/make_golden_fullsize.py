#!/usr/bin/env python3
"""Golden digests at BASELINE.json's full sizes, produced by IMPORTING THE REFERENCE (build container only; the
reference needs minutes per frame pair at these sizes -- 1080p pyramidal: ~4 min, 4K: ~15 min on one core):
  configs[1]  640x480    single-scale 5x5          (lucas_kanade_core.lucas_kanade_single_scale)
  configs[2]  1920x1080  3 levels, 5x5, 3 iterations (lucas_kanade_pyramidal.lucas_kanade_pyramidal)
  configs[3]  3840x2160  one pair of the 64-pair job, same parameters
  configs[4]  7680x4320  single-scale 7x7 in the reference's own fp32 (what the opt-in fp16 mode is measured against)
on the bench workload's synthetic frames (optical-flow-fpga_amd/python/oflk_synth.py, pair_index 0); plus
  m1 .. m7   mid-size cases with other levels / iterations / windows and odd shapes (5 - 70 s each)
  e1 .. e3   small motions (identical frames; smooth frames shifted by a fraction of a pixel) whose levels leave the
             iteration loop early: the iteration counts are read from the lines the reference prints
Usage: python3 tests/golden/make_golden_fullsize.py [c1 .. c4] [m1 .. m7] [e1 .. e3]   (default: c1 c2)
       GOLD_OUT=/tmp/x.json ... writes to another file (a second process running beside the first; merge by hand)
Output: tests/golden/reference_fullsize.json (entries are merged into the existing file)."""
import contextlib
import hashlib
import io
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/python")
import lucas_kanade_core as R_core  # noqa: E402  (reference)
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)

R_pyr.visualize_pyramid_level = lambda *a, **k: None

CASES = {
    "c1": {"shape": [480, 640], "mode": "single_scale", "window_size": 5},
    "c2": {"shape": [1080, 1920], "mode": "pyramidal", "levels": 3, "window_size": 5, "iterations": 3},
    "c3": {"shape": [2160, 3840], "mode": "pyramidal", "levels": 3, "window_size": 5, "iterations": 3},
    "c4": {"shape": [4320, 7680], "mode": "single_scale", "window_size": 7},
    # other parameters and odd shapes at a size the reference finishes in a minute (pair_index = another seed)
    "m1": {"shape": [480, 640], "mode": "pyramidal", "levels": 4, "window_size": 5, "iterations": 5, "pair_index": 1},
    "m2": {"shape": [480, 640], "mode": "pyramidal", "levels": 3, "window_size": 7, "iterations": 2, "pair_index": 2},
    "m3": {"shape": [480, 640], "mode": "pyramidal", "levels": 2, "window_size": 9, "iterations": 3, "pair_index": 3},
    "m4": {"shape": [480, 640], "mode": "pyramidal", "levels": 3, "window_size": 11, "iterations": 1, "pair_index": 4},
    "m5": {"shape": [481, 643], "mode": "pyramidal", "levels": 3, "window_size": 5, "iterations": 3, "pair_index": 5},
    "m6": {"shape": [360, 1001], "mode": "pyramidal", "levels": 3, "window_size": 3, "iterations": 3, "pair_index": 6},
    "m7": {"shape": [555, 333], "mode": "single_scale", "window_size": 9, "pair_index": 7},
    # small motions: levels leave their iteration loop early (lucas_kanade_pyramidal.py:221-223); identical frames
    "e1": {"shape": [480, 640], "mode": "pyramidal", "levels": 3, "window_size": 5, "iterations": 4, "pair_index": 8, "dx": 0.0, "dy": 0.0},
    "e2": {"shape": [480, 640], "mode": "pyramidal", "levels": 3, "window_size": 5, "iterations": 4, "pair_index": 9, "dx": 0.04, "dy": -0.02, "smooth": True},
    "e3": {"shape": [480, 640], "mode": "pyramidal", "levels": 3, "window_size": 5, "iterations": 4, "pair_index": 10, "dx": 0.3, "dy": 0.1, "smooth": True},
}


def digest(a):
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


def synth(h, w, pair_index=0, dx=3.0, dy=-1.5, smooth=False):
    # the product's generator, loaded by path so that the reference's modules (same names as the shims) stay the ones imported above
    import importlib.util

    spec = importlib.util.spec_from_file_location("oflk_synth", HERE.parents[1] / "optical-flow-fpga_amd" / "python" / "oflk_synth.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return (m.synth_pair_smooth if smooth else m.synth_pair)(h, w, pair_index, dx, dy)


def main():
    want = sys.argv[1:] or ["c1", "c2"]
    import os

    path = Path(os.environ.get("GOLD_OUT", HERE / "reference_fullsize.json"))   # (a second process writes elsewhere; merge by hand)
    out = json.loads(path.read_text()) if path.exists() else {}
    for key in want:
        c = dict(CASES[key])
        h, w = c["shape"]
        c.setdefault("pair_index", 0)
        p, q = synth(h, w, c["pair_index"], c.get("dx", 3.0), c.get("dy", -1.5), c.get("smooth", False))
        t0 = time.time()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            if c["mode"] == "single_scale":
                u, v = R_core.lucas_kanade_single_scale(p, q, c["window_size"])
            else:
                u, v = R_pyr.lucas_kanade_pyramidal(p, q, c["levels"], c["window_size"], c["iterations"])
        c.update(u_sha256=digest(u), v_sha256=digest(v), mean_abs_u=float(np.abs(u).mean(dtype=np.float64)),
                 mean_abs_v=float(np.abs(v).mean(dtype=np.float64)), reference_seconds=round(time.time() - t0, 1))
        if c["mode"] == "pyramidal":
            # iterations the reference ran per level, counted from the residual lines it prints (coarse -> fine)
            runs = []
            for line in buf.getvalue().splitlines():
                if line.startswith("Processing pyramid level"):
                    runs.append(0)
                elif line.startswith("  Iteration ") and runs:
                    runs[-1] += 1
            c["iters_run"] = runs
        out[key] = c
        print(key, c, flush=True)
        path.write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

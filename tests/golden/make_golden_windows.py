#!/usr/bin/env python3
"""Golden digests for the windows outside 3x3 ... 11x11 (the reference takes any window_size,
python/lucas_kanade_core.py:104-119): 1x1 (window_size 1), 13x13 (12, 13), 15x15, 21x21 -- a 13x13 window is 169 products,
beyond NumPy's 128-element pairwise block, so np.sum's order changes shape there.  Produced by IMPORTING THE REFERENCE
(build container only) on a 96 x 128 crop of two of the 13 patterns.  Output: tests/golden/reference_windows.json."""
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
CROP = (slice(60, 156), slice(80, 208))
SINGLE = [1, 12, 13, 15, 21]
PYRAMIDAL = [(1, 2, 2), (13, 2, 2)]   # window_size, levels, iterations
PATTERNS = ["translate_medium", "rotate_small"]


def digest(a):
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    z = np.load(HERE / "patterns_320x240.npz")
    out = {"crop": [[CROP[0].start, CROP[0].stop], [CROP[1].start, CROP[1].stop]], "patterns": {}}
    t0 = time.time()
    for name in PATTERNS:
        p = np.ascontiguousarray(z["frame_0"].astype(np.float32)[CROP])
        c = np.ascontiguousarray(z[f"frame_1__{name}"].astype(np.float32)[CROP])
        e = out["patterns"][name] = {"single_scale": {}, "pyramidal": {}}
        for win in SINGLE:
            u, v = R_core.lucas_kanade_single_scale(p, c, win)
            e["single_scale"][str(win)] = {"u_sha256": digest(u), "v_sha256": digest(v), "nonzero_u": int(np.count_nonzero(u))}
            print(name, "single", win, f"{time.time() - t0:.0f}s", flush=True)
        for win, levels, iters in PYRAMIDAL:
            with contextlib.redirect_stdout(io.StringIO()):
                u, v = R_pyr.lucas_kanade_pyramidal(p, c, levels, win, iters)
            e["pyramidal"][str(win)] = {"levels": levels, "iterations": iters, "u_sha256": digest(u), "v_sha256": digest(v)}
            print(name, "pyramidal", win, f"{time.time() - t0:.0f}s", flush=True)
    (HERE / "reference_windows.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

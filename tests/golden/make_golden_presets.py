#!/usr/bin/env python3
"""Golden digests for the reference's other pyramid presets (python/verification_config.yaml:78-103:
shallow = 2 levels, deep = 4 levels, large_window = 7x7) on four of the 13 patterns, produced by
IMPORTING THE REFERENCE (build container only).  Output: tests/golden/reference_presets.json."""
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
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)

R_pyr.visualize_pyramid_level = lambda *a, **k: None

PRESETS = {"shallow": (2, 5, 3), "deep": (4, 5, 3), "large_window": (3, 7, 3)}
PATTERNS = ["translate_medium", "rotate_small", "translate_extreme", "no_motion"]


def digest(a):
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    z = np.load(HERE / "patterns_320x240.npz")
    out = {}
    t0 = time.time()
    for name in PATTERNS:
        p, c = z["frame_0"].astype(np.float32), z[f"frame_1__{name}"].astype(np.float32)
        out[name] = {}
        for preset, (levels, win, iters) in PRESETS.items():
            with contextlib.redirect_stdout(io.StringIO()):
                u, v = R_pyr.lucas_kanade_pyramidal(p, c, levels, win, iters)
            out[name][preset] = {"levels": levels, "window_size": win, "iterations": iters,
                                 "u_sha256": digest(u), "v_sha256": digest(v)}
            print(name, preset, f"{time.time() - t0:.0f}s", flush=True)
    (HERE / "reference_presets.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

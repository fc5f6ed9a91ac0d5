#!/usr/bin/env python3
"""Golden stdout of the reference's lucas_kanade_pyramidal (the progress lines of
python/lucas_kanade_pyramidal.py:172-222 that README.md:258-299 documents), by IMPORTING THE
REFERENCE.  Build container only; the output file is data (captured text), the reference never
travels.  Two patterns: translate_small (every level runs its three iterations) and no_motion (every
level converges after one: the "Converged after" line).

Usage:  python tests/golden/make_golden_stdout.py      (about 30 s)   -> tests/golden/reference_stdout.json
"""
from __future__ import annotations

import contextlib
import io
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/python")
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)

R_pyr.visualize_pyramid_level = lambda *a, **k: None   # the PNG side effect (:226) is not wanted here


def main() -> None:
    z = np.load(HERE / "patterns_320x240.npz")
    out = {"args": {"num_levels": 3, "window_size": 5, "num_iterations": 3}, "stdout": {}}
    for name in ("translate_small", "no_motion"):
        p, c = z["frame_0"].astype(np.float32), z[f"frame_1__{name}"].astype(np.float32)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            R_pyr.lucas_kanade_pyramidal(p, c, 3, 5, 3)
        out["stdout"][name] = buf.getvalue()
        print(name, "->", len(buf.getvalue().splitlines()), "lines")
    (HERE / "reference_stdout.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

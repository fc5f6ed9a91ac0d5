#!/usr/bin/env python3
"""Golden pyramids for scale factors other than the reference's default 0.5, by IMPORTING THE REFERENCE
(build container only; the output is data).  build_gaussian_pyramid(image, num_levels, scale_factor)
of python/lucas_kanade_pyramidal.py:23-63 with sigma = 1 / scale_factor: the Gaussian weights then come
from NumPy's exp, which the C oracle and the HIP library replace by libm's exp (only the sigma = 2 table
is embedded) -- this fixture measures what that costs.

Round 4: a second file with nine further scale factors (pyramid_scales_more.npz), for the entry point that takes the
weights from the caller (oflk_build_pyramid_w: the shim computes them with NumPy, exactly as SciPy does) -- with it every
scale factor gives the reference's pyramid, and these fixtures check that on values nobody tuned for.

Usage:  python tests/golden/make_golden_scales.py      -> tests/golden/pyramid_scales.npz, pyramid_scales_more.npz
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/python")
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)


def main():
    rng = np.random.default_rng(20260220)
    img = rng.integers(0, 256, (61, 83)).astype(np.float32)
    out = {"image": img}
    for sf in (0.6, 0.4, 0.75, 0.3):
        pyr = R_pyr.build_gaussian_pyramid(img, 3, scale_factor=sf)
        for l, a in enumerate(pyr):
            out[f"sf{sf}_level{l}"] = a.astype(np.float32)
        print(sf, [a.shape for a in pyr])
    np.savez_compressed(HERE / "pyramid_scales.npz", **out)
    img2 = rng.integers(0, 256, (70, 97)).astype(np.float32)
    more = {"image": img2}
    for sf in (0.35, 0.45, 0.55, 0.65, 0.8, 0.9, 0.25, 0.7, 1.0 / 3.0):
        levels = 3 if sf >= 0.3 else 2
        pyr = R_pyr.build_gaussian_pyramid(img2, levels, scale_factor=sf)
        for l, a in enumerate(pyr):
            more[f"sf{sf!r}_level{l}"] = a.astype(np.float32)
        print(sf, [a.shape for a in pyr])
    np.savez_compressed(HERE / "pyramid_scales_more.npz", **more)


if __name__ == "__main__":
    main()

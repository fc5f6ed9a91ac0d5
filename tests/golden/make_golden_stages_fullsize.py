#!/usr/bin/env python3
"""Digests of the reference's STAGE functions at 1920x1080 and 3840x2160 (they are SciPy calls: seconds), produced by
IMPORTING THE REFERENCE (build container only): compute_gradients, build_gaussian_pyramid (3 levels), warp_image and
upsample_flow on the bench workload's frames (oflk_synth.synth_pair) and a synthetic flow field (oflk_synth.synth_flow).
Output: tests/golden/reference_stages_fullsize.json."""
import hashlib
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/python")
import lucas_kanade_core as R_core  # noqa: E402  (reference)
import lucas_kanade_pyramidal as R_pyr  # noqa: E402  (reference)

spec = importlib.util.spec_from_file_location("oflk_synth", HERE.parents[1] / "optical-flow-fpga_amd" / "python" / "oflk_synth.py")
S = importlib.util.module_from_spec(spec)
spec.loader.exec_module(S)


def digest(a):
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    out = {}
    for key, (h, w) in {"1080p": (1080, 1920), "4k": (2160, 3840), "odd": (1081, 1923)}.items():
        p, c = S.synth_pair(h, w, pair_index=0)
        ix, iy, it = R_core.compute_gradients(p, c)
        pyr = R_pyr.build_gaussian_pyramid(c, 3)
        fu, fv = S.synth_flow(h, w, seed=1)
        warped = R_pyr.warp_image(c, fu, fv)
        hc, wc = pyr[1].shape
        cu, cv = S.synth_flow(hc, wc, seed=2)
        uu, uv = R_pyr.upsample_flow(cu, cv, (h, w))
        out[key] = {"shape": [h, w], "gradients": [digest(ix), digest(iy), digest(it)],
                    "pyramid_shapes": [list(a.shape) for a in pyr], "pyramid": [digest(a) for a in pyr],
                    "warp": digest(warped), "upsample_from": [hc, wc], "upsample": [digest(uu), digest(uv)]}
        print(key, out[key]["pyramid_shapes"], flush=True)
    (HERE / "reference_stages_fullsize.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

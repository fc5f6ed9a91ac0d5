#!/usr/bin/env python3
"""Which relaxed stage costs how much EPE?  (CPU experiment; oracle = exact side.)"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools" / "experiments"))
import oflk_oracle as O
import fast_mode_emulation as F
f32 = np.float32

def pyr(prev, curr, fast_pyr, fast_warp, fast_sums, fast_up, L=3, win=5, K=3):
    w = np.asarray(O.gaussian_kernel1d(2.0))
    def build(img):
        if not fast_pyr:
            return O.build_gaussian_pyramid(img, L, 0.5)
        out = [img]
        for _ in range(L - 1):
            a = out[0]
            out.insert(0, F.resample_fast(F.gauss_fast(a, w), int(a.shape[0] * 0.5), int(a.shape[1] * 0.5)))
        return out
    pp, pc = build(prev), build(curr)
    fu = np.zeros(pp[0].shape, f32); fv = np.zeros(pp[0].shape, f32)
    for l in range(L):
        if l > 0:
            Ht, Wt = pp[l].shape; Hc, Wc = fu.shape
            if fast_up:
                fu, fv = (F.resample_fast(fu, Ht, Wt) * f32(Wt / Wc)).astype(f32), (F.resample_fast(fv, Ht, Wt) * f32(Ht / Hc)).astype(f32)
            else:
                fu, fv = O.upsample_flow(fu, fv, (Ht, Wt))
        for _ in range(K):
            wc = F.warp_fast(pc[l], fu, fv) if fast_warp else O.warp_image(pc[l], fu, fv)
            du, dv = F.lk_fast(pp[l], wc, win) if fast_sums else O.lucas_kanade_single_scale(pp[l], wc, win)
            fu = (fu + du).astype(f32); fv = (fv + dv).astype(f32)
            if np.mean(np.abs(du)) < 0.01 and np.mean(np.abs(dv)) < 0.01:
                break
    return fu, fv

z = np.load(ROOT / "tests/golden/patterns_320x240.npz")
f0 = z["frame_0"].astype(f32)
cfgs = {"none": (0,0,0,0), "pyr": (1,0,0,0), "warp": (0,1,0,0), "sums": (0,0,1,0), "up": (0,0,0,1), "all": (1,1,1,1)}
print("pattern".ljust(20) + "".join(k.rjust(11) for k in cfgs))
for k in z.files:
    if not k.startswith("frame_1__"): continue
    f1 = z[k].astype(f32)
    pu, pv = O.lucas_kanade_pyramidal(f0, f1, 3, 5, 3)[:2]
    row = k[9:].ljust(20)
    for name, c in cfgs.items():
        qu, qv = pyr(f0, f1, *c)
        row += f"{F.epe(pu, pv, qu, qv):11.2e}"
    print(row, flush=True)

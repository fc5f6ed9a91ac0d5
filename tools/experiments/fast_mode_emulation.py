#!/usr/bin/env python3
"""Experiment (CPU, numpy): how far does an all-fp32, free-summation-order variant of the
path drift from the bit-exact one on the 13 reference patterns?  Decides whether a "fast"
device mode can meet the north-star tolerance (EPE <= 1e-4 vs the Python reference).
Uses the oracle as the exact side, so this is a test-side tool only."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import oflk_oracle as O  # noqa: E402

f32 = np.float32


def box5(a, hw):
    """separable fp32 window sum, left-to-right order"""
    H, W = a.shape
    n = 2 * hw + 1
    r = np.zeros((H, W - 2 * hw), f32)
    for k in range(n):
        r = (r + a[:, k:W - 2 * hw + k]).astype(f32)
    c = np.zeros((H - 2 * hw, W - 2 * hw), f32)
    for k in range(n):
        c = (c + r[k:H - 2 * hw + k, :]).astype(f32)
    return c


def lk_fast(prev, curr, win=5):
    Ix, Iy, It = O.compute_gradients(prev, curr)   # Sobel is cheap: keep it exact
    hw = win // 2
    H, W = prev.shape
    u = np.zeros((H, W), f32)
    v = np.zeros((H, W), f32)
    if H <= 2 * hw or W <= 2 * hw:
        return u, v
    Sxx = box5((Ix * Ix).astype(f32), hw)
    Syy = box5((Iy * Iy).astype(f32), hw)
    Sxy = box5((Ix * Iy).astype(f32), hw)
    Sxt = box5((Ix * It).astype(f32), hw)
    Syt = box5((Iy * It).astype(f32), hw)
    det = (Sxx * Syy - Sxy * Sxy).astype(f32)
    ok = np.abs(det) > f32(1e-4)
    b0, b1 = -Sxt, -Syt
    with np.errstate(all="ignore"):
        uu = ((Syy * b0 - Sxy * b1) / det).astype(f32)
        vv = ((Sxx * b1 - Sxy * b0) / det).astype(f32)
    u[hw:H - hw, hw:W - hw] = np.where(ok, uu, 0)
    v[hw:H - hw, hw:W - hw] = np.where(ok, vv, 0)
    return u, v


def bilinear_f32(img, y0, x0, fy, fx, inside):
    H, W = img.shape
    y0c = np.clip(y0, 0, H - 1)
    x0c = np.clip(x0, 0, W - 1)
    y1c = np.clip(y0 + 1, 0, H - 1)
    x1c = np.clip(x0 + 1, 0, W - 1)
    wy1, wx1 = fy.astype(f32), fx.astype(f32)
    wy0, wx0 = (f32(1) - wy1).astype(f32), (f32(1) - wx1).astype(f32)
    top = (img[y0c, x0c] * wx0 + img[y0c, x1c] * wx1).astype(f32)
    bot = (img[y1c, x0c] * wx0 + img[y1c, x1c] * wx1).astype(f32)
    r = (top * wy0 + bot * wy1).astype(f32)
    return np.where(inside, r, f32(0)).astype(f32)


def warp_fast(img, fu, fv):
    H, W = img.shape
    gy, gx = np.mgrid[0:H, 0:W]
    flv, flu = np.floor(fv), np.floor(fu)
    fy, fx = (fv - flv).astype(f32), (fu - flu).astype(f32)
    y0, x0 = gy + flv.astype(np.int64), gx + flu.astype(np.int64)
    y = y0 + fy.astype(np.float64)
    x = x0 + fx.astype(np.float64)
    inside = (y >= 0) & (y <= H - 1) & (x >= 0) & (x <= W - 1)
    return bilinear_f32(img, y0, x0, fy, fx, inside)


def resample_fast(img, Ho, Wo):
    H, W = img.shape
    ys = np.linspace(0, H - 1, Ho)
    xs = np.linspace(0, W - 1, Wo)
    y, x = np.meshgrid(ys, xs, indexing="ij")
    y0, x0 = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
    fy, fx = (y - y0).astype(f32), (x - x0).astype(f32)
    inside = np.ones(y.shape, bool)
    return bilinear_f32(img, y0, x0, fy, fx, inside)


def gauss_fast(img, w):
    """fp32 separable 17-tap, symmetric pairing like SciPy but fp32 throughout"""
    w = np.concatenate([w[:0:-1], w]).astype(f32)   # half kernel (distance 0..r) -> full
    r = (len(w) - 1) // 2
    def axis0(a):
        p = np.pad(a, ((r, r), (0, 0)), mode="symmetric")
        H = a.shape[0]
        t = (p[r:r + H] * w[r]).astype(f32)
        for k in range(1, r + 1):
            t = (t + (p[r - k:r - k + H] + p[r + k:r + k + H]).astype(f32) * w[r + k]).astype(f32)
        return t
    return axis0(axis0(img).T.copy()).T.copy()


def pyr_fast(prev, curr, L=3, win=5, K=3):
    w = O.gaussian_kernel1d(2.0)
    def build(img):
        out = [img]
        for _ in range(L - 1):
            a = out[0]
            b = gauss_fast(a, np.asarray(w))
            out.insert(0, resample_fast(b, int(a.shape[0] * 0.5), int(a.shape[1] * 0.5)))
        return out
    pp, pc = build(prev), build(curr)
    fu = np.zeros(pp[0].shape, f32)
    fv = np.zeros(pp[0].shape, f32)
    for l in range(L):
        if l > 0:
            Ht, Wt = pp[l].shape
            Hc, Wc = fu.shape
            fu = (resample_fast(fu, Ht, Wt) * f32(Wt / Wc)).astype(f32)
            fv = (resample_fast(fv, Ht, Wt) * f32(Ht / Hc)).astype(f32)
        for _ in range(K):
            wc = warp_fast(pc[l], fu, fv)
            du, dv = lk_fast(pp[l], wc, win)
            fu = (fu + du).astype(f32)
            fv = (fv + dv).astype(f32)
            if np.mean(np.abs(du)) < 0.01 and np.mean(np.abs(dv)) < 0.01:
                break
    return fu, fv


def epe(a, b, c, d):
    return float(np.mean(np.sqrt((a - c).astype(np.float64) ** 2 + (b - d).astype(np.float64) ** 2)))


def main():
    z = np.load(ROOT / "tests/golden/patterns_320x240.npz")
    f0 = z["frame_0"].astype(f32)
    print(f"{'pattern':22s} {'single EPE':>12s} {'max':>10s} {'pyr EPE':>12s} {'max':>10s}")
    for k in z.files:
        if not k.startswith("frame_1__"):
            continue
        f1 = z[k].astype(f32)
        eu, ev = O.lucas_kanade_single_scale(f0, f1, 5)
        fu, fv = lk_fast(f0, f1, 5)
        pu, pv = O.lucas_kanade_pyramidal(f0, f1, 3, 5, 3)[:2]
        qu, qv = pyr_fast(f0, f1)
        m1 = float(np.max(np.hypot(eu - fu, ev - fv)))
        m2 = float(np.max(np.hypot(pu - qu, pv - qv)))
        print(f"{k[9:]:22s} {epe(eu, ev, fu, fv):12.3e} {m1:10.3e} {epe(pu, pv, qu, qv):12.3e} {m2:10.3e}")




def stage_check():
    z = np.load(ROOT / "tests/golden/patterns_320x240.npz")
    f0 = z["frame_0"].astype(f32)
    f1 = z["frame_1__translate_medium"].astype(f32)
    w = np.asarray(O.gaussian_kernel1d(2.0))
    g_e = O.gaussian_filter(f0, 2.0)
    g_f = gauss_fast(f0, w)
    print("gauss max diff", np.abs(g_e - g_f).max())
    pe = O.build_gaussian_pyramid(f0, 3, 0.5)
    r_f = resample_fast(g_e, 120, 160)
    print("resample max diff", np.abs(pe[1] - r_f).max())
    fu = (np.random.default_rng(0).standard_normal(f0.shape) * 2).astype(f32)
    fv = (np.random.default_rng(1).standard_normal(f0.shape) * 2).astype(f32)
    print("warp max diff", np.abs(O.warp_image(f0, fu, fv) - warp_fast(f0, fu, fv)).max())
    cu, cv = fu[:120, :160].copy(), fv[:120, :160].copy()
    eu, ev = O.upsample_flow(cu, cv, (240, 320))
    print("upsample max diff", np.abs(eu - (resample_fast(cu, 240, 320) * f32(2)).astype(f32)).max())
    wc = O.warp_image(f1, fu * 0, fv * 0)
    a, b = O.lucas_kanade_single_scale(g_e, O.gaussian_filter(f1, 2.0), 5)
    c, d = lk_fast(g_e, O.gaussian_filter(f1, 2.0), 5)
    print("lk on blurred frames: EPE", epe(a, b, c, d), "max", np.max(np.hypot(a - c, b - d)), "max |u| exact", np.abs(a).max())


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stages":
    stage_check()


if __name__ == "__main__" and len(sys.argv) == 1:
    main()

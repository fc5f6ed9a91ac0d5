"""Deterministic synthetic frame pairs for benchmarks and parity tests.

Resolution-scaled version of the sinusoid texture the reference uses for its RTL
test frames (generate_test_frames_natural.py:49-64) plus band-limited noise, so
that nearly every window is textured.  Values are
integer-valued float32 in [0, 255], exactly what the verifier feeds the hot path
(optical_flow_verifier.py:61-65).  Host-side input generation only.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

BASE_SEED = 20260220


def synth_pair(H: int, W: int, pair_index: int = 0, dx: float = 3.0, dy: float = -1.5
               ) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (frame_prev, frame_curr) float32 [H, W]; curr = prev shifted by (dx, dy)."""
    from scipy.ndimage import gaussian_filter, shift

    x = np.linspace(0.0, 4.0 * np.pi * W / 320.0, W)
    y = np.linspace(0.0, 3.0 * np.pi * H / 240.0, H)
    X, Y = np.meshgrid(x, y)
    rng = np.random.default_rng(BASE_SEED + int(pair_index))
    noise = gaussian_filter(rng.standard_normal((H, W)), sigma=1.5)
    noise /= max(float(noise.std()), 1e-12)
    p = (128.0 + 50.0 * np.sin(X) * np.cos(Y) + 30.0 * np.cos(2 * X + 0.5) * np.sin(1.5 * Y)
         + 20.0 * np.sin(3 * X - 0.3) * np.cos(2.5 * Y + 0.7) + 12.0 * noise)
    f0 = np.clip(p, 0, 255).astype(np.uint8)
    f1 = shift(f0.astype(np.float64), (dy, dx), order=1, mode="constant", cval=128.0)
    f1 = np.clip(f1, 0, 255).astype(np.uint8)
    return f0.astype(np.float32), f1.astype(np.float32)


def synth_flow(H: int, W: int, seed: int = 0, amplitude: float = 3.0) -> Tuple[np.ndarray, np.ndarray]:
    """A smooth float32 flow field (u, v) with sub-pixel noise, for tests of warp_image / upsample_flow at any size:
    generic fractions, a few samples that leave the image at the borders."""
    rng = np.random.default_rng(BASE_SEED + 7919 * (int(seed) + 1))
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    u = amplitude * np.sin(x / 50.0) * np.cos(y / 70.0) + rng.normal(0.0, 0.3, (H, W))
    v = amplitude * 0.5 * np.cos(x / 35.0 + 0.4) * np.sin(y / 45.0) + rng.normal(0.0, 0.3, (H, W))
    return u.astype(np.float32), v.astype(np.float32)


def synth_pair_smooth(H: int, W: int, pair_index: int = 0, dx: float = 0.3, dy: float = 0.1
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """Noise-free float32 frames (not 8-bit): the texture of synth_pair blurred, curr = a cubic sub-pixel shift of it.
    Small motions converge within an iteration or two, so pyramid levels leave their loop early
    (lucas_kanade_pyramidal.py:221-223) at different iteration counts."""
    from scipy.ndimage import gaussian_filter, shift

    p, _ = synth_pair(H, W, pair_index)
    p = gaussian_filter(p.astype(np.float64), 2.0)
    c = shift(p, (dy, dx), order=3, mode="nearest")
    return p.astype(np.float32), c.astype(np.float32)

#!/usr/bin/env python3
"""Host-to-host timings through the drop-in Python modules (numpy in, numpy out): what a caller
of the reference's functions sees, PCIe copies included.  Development tool; the bench figure
(`bench.py`) is measured with inputs resident in HBM."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))
os.environ.setdefault("OFLK_QUIET", "1")

import lucas_kanade_core as K  # noqa: E402
import lucas_kanade_pyramidal as P  # noqa: E402
from oflk_synth import synth_pair  # noqa: E402


def timed(fn, reps):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def batch_main(B=32, h=1080, w=1920):
    """a batch of B pairs through the C ABI, host arrays in and out: large batches are cut into chunks whose H2D, kernels
    and D2H overlap (csrc/oflk.hip run_batch_chunked)"""
    import ctypes

    import _oflk

    L = _oflk.lib()
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    pairs = [synth_pair(h, w, pair_index=i) for i in range(4)]
    a = np.ascontiguousarray(np.stack([pairs[i % 4][0] for i in range(B)]))
    b = np.ascontiguousarray(np.stack([pairs[i % 4][1] for i in range(B)]))
    u, v = np.empty_like(a), np.empty_like(a)
    log, runs = np.zeros((B, 3, 3, 2), np.float32), np.zeros((B, 3), np.int32)
    tf = timed(lambda: _oflk.check(L.oflk_pyramidal_batch(a.ctypes.data_as(f32p), b.ctypes.data_as(f32p), B, h, w, 3, 5, 3,
                                                          u.ctypes.data_as(f32p), v.ctypes.data_as(f32p), log.ctypes.data_as(f32p),
                                                          runs.ctypes.data_as(i32p))), 5)
    a8, b8 = a.astype(np.uint8), b.astype(np.uint8)
    t8 = timed(lambda: _oflk.check(L.oflk_pyramidal_u8(a8.ctypes.data, b8.ctypes.data, B, h, w, 3, 5, 3, u.ctypes.data_as(f32p),
                                                       v.ctypes.data_as(f32p), log.ctypes.data_as(f32p), runs.ctypes.data_as(i32p))), 5)
    print(f"{B} x {w}x{h} pyramidal, one call, host to host: float32 frames {tf / B * 1e3:.3f} ms/pair ({B * h * w / tf / 1e6:.0f} Mpix/s, "
          f"{a.nbytes * 4 / tf / 1e9:.1f} GB/s over PCIe), uint8 frames {t8 / B * 1e3:.3f} ms/pair ({B * h * w / t8 / 1e6:.0f} Mpix/s)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "batch":
        batch_main(*[int(x) for x in sys.argv[2:]])
        return
    for (h, w) in ((480, 640), (1080, 1920), (2160, 3840)):
        a, b = synth_pair(h, w, pair_index=1)
        ts = timed(lambda: K.lucas_kanade_single_scale(a, b, 5), 20)
        tp = timed(lambda: P.lucas_kanade_pyramidal(a, b, 3, 5, 3), 20)
        mb = a.nbytes * 4 / 1e6
        print(f"{w}x{h}: single-scale {ts*1e3:7.3f} ms ({h*w/ts/1e6:8.1f} Mpix/s)   pyramidal {tp*1e3:7.3f} ms "
              f"({h*w/tp/1e6:8.1f} Mpix/s)   [{mb:.1f} MB over PCIe per call]")


if __name__ == "__main__":
    main()

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


def main():
    for (h, w) in ((480, 640), (1080, 1920), (2160, 3840)):
        a, b = synth_pair(h, w, pair_index=1)
        ts = timed(lambda: K.lucas_kanade_single_scale(a, b, 5), 20)
        tp = timed(lambda: P.lucas_kanade_pyramidal(a, b, 3, 5, 3), 20)
        mb = a.nbytes * 4 / 1e6
        print(f"{w}x{h}: single-scale {ts*1e3:7.3f} ms ({h*w/ts/1e6:8.1f} Mpix/s)   pyramidal {tp*1e3:7.3f} ms "
              f"({h*w/tp/1e6:8.1f} Mpix/s)   [{mb:.1f} MB over PCIe per call]")


if __name__ == "__main__":
    main()

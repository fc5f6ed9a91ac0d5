#!/usr/bin/env python3
"""Device-memory plateau check of the host entry points (development tool, run on the GPU box): many calls over more shapes
than the plan cache holds (six), float32 and uint8, single pairs and chunked batches; the free device memory after the
first sweep and after the last one must agree (plans are evicted and freed, staging buffers only grow to the largest shape).
Usage: python3 tools/leak_check.py [sweeps]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import torch  # first: one HIP runtime in the process

    import lucas_kanade_core as K
    import lucas_kanade_pyramidal as P
    from oflk_synth import synth_pair

    shapes = [(120, 160), (240, 320), (241, 323), (480, 640), (97, 1031), (600, 800), (333, 555), (1080, 1920), (64, 64), (31, 47)]
    pairs = {s: synth_pair(*s, pair_index=i) for i, s in enumerate(shapes)}
    free = []
    for sweep in range(sweeps):
        for s in shapes:
            p, c = pairs[s]
            K.lucas_kanade_single_scale(p, c, 5)
            P.lucas_kanade_pyramidal(p, c, 3 if min(s) >= 64 else 2, 5, 2)
            if sweep % 4 == 0:
                K.lucas_kanade_single_scale(p.astype(np.uint8), c.astype(np.uint8), 7)
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
        if sweep in (0, sweeps - 1) or sweep % 10 == 0:
            print(f"sweep {sweep}: free device memory {free[-1] / 2**20:.1f} MiB", flush=True)
    drift = free[1] - free[-1]
    print(f"free after sweep 1: {free[1] / 2**20:.1f} MiB, after sweep {sweeps - 1}: {free[-1] / 2**20:.1f} MiB, drift {drift / 2**20:.2f} MiB")
    sys.exit(0 if abs(drift) <= 8 * 2**20 else 1)


if __name__ == "__main__":
    main()

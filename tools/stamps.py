#!/usr/bin/env python3
"""In-kernel timeline of the finest-level fused LK iteration (development tool).

Needs the diagnostic build (never the shipped library):
    hipcc ... -DOFLK_STAMPS -DOFLK_STAMP_MASK=0x7ff9 -shared -o tools/liboflk_stamps.so csrc/oflk.hip
    OFLK_LIB=tools/liboflk_stamps.so python3 tools/stamps.py [out.json]

Every wave records s_memtime at fixed points of each tile it processes (k_lkw, OFLK_STAMP);
this script turns the raw stamps of the last finest-level launch into the mean cycles a wave
spends per section and tile, and prints one block's timeline.  The diagnostic build fences the
scheduler at every stamp, so read SHARES, not the absolute length.
"""
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))

# section = interval ENDING at stamp i (stamps 1 and 2 are compiled out: they cost 17 VGPRs)
SECTION = {
    3: "stage 1a: carry->LDS, coalesced loads (prev,u,v), wait, taps, gathers of the first batch (a continuing tile: all 7 cells), wait, fp64 sums",
    4: "stage 1b: taps of the second batch, gathers issued (first tile of a segment only: 9 cells = 5 + 4)",
    5: "stage 1c: wait for those gathers, fp64 sums (first tile of a segment only)",
    6: "stage 1d: avg / It -> LDS",
    7: "barrier 1",
    8: "stage 2: Sobel from LDS (+ carry rows)",
    9: "barrier 2",
    10: "products -> LDS",
    11: "barrier 3",
    12: "stage 3: window sums (LDS reads + adds)",
    13: "solve, flow += d, stores, |d| wave reduction",
    14: "barrier 4",
    0: "loop top (addresses of the next tile)",
}
ORDER = [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]


def main():
    import torch

    import _oflk
    from oflk_synth import synth_pair

    L = _oflk.lib()
    L.oflk_debug_stamps.restype = ctypes.c_long
    L.oflk_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    dev = torch.device("cuda", 0)
    B, H, W = 32, 1080, 1920
    host = [synth_pair(H, W, i) for i in range(4)]
    prev = torch.stack([torch.from_numpy(host[b % 4][0]) for b in range(B)]).to(dev)
    curr = torch.stack([torch.from_numpy(host[b % 4][1]) for b in range(B)]).to(dev)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    plan = _oflk.Plan(0, B, H, W, 3, 5, 3)
    st = torch.cuda.current_stream().cuda_stream
    plan.set_profiling(2)
    for _ in range(5):
        plan.pyramidal(prev.data_ptr(), curr.data_ptr(), u.data_ptr(), v.data_ptr(), st)
    torch.cuda.synchronize()
    kt = plan.kernel_times().get("lk_iter_finest", {"total_ms": 0.0, "launches": 1})
    launch_us = 1e3 * kt["total_ms"] / max(kt["launches"], 1)
    print(f"diagnostic build: finest-level iteration launch {launch_us:.1f} us (the product build is faster: no fences)")
    nblk = L.oflk_debug_stamps(plan._h, None, 0)
    raw = np.zeros((nblk, 4, 8, 16), np.uint32)
    L.oflk_debug_stamps(plan._h, raw.ctypes.data_as(ctypes.c_void_p), nblk)
    if len(sys.argv) > 2:   # the raw stamps (17 MB) only on request
        np.save(Path(sys.argv[1]).with_suffix(".raw.npy"), raw)
    ntile = raw[:, :, 0, 15].astype(int)              # tiles the block walked (0 = block exited at once)
    sums = {i: 0.0 for i in ORDER}
    count = 0
    first_tile = {i: 0.0 for i in ORDER}
    n_first = 0
    for b in range(nblk):
        n = ntile[b, 0]
        if n == 0:
            continue
        for w in range(4):
            for t in range(n):
                s = raw[b, w, t].astype(np.int64)
                prev_t = None
                for i in ORDER:
                    if i == 0:
                        # from the previous tile's last stamp (14) to this tile's top
                        if t > 0:
                            sums[0] += (int(s[0]) - int(raw[b, w, t - 1, 14])) & 0xffffffff
                    else:
                        d = (int(s[i]) - int(prev_t)) & 0xffffffff
                        if d > 1000000:   # a stamp this tile did not pass (a continuing tile has ONE gather batch: no 4 / 5)
                            continue
                        if t == 0:
                            first_tile[i] += d
                        else:
                            sums[i] += d
                    prev_t = s[i]
                if t == 0:
                    n_first += 1
                else:
                    count += 1
    tot = sum(sums.values())
    out = {"workload": f"{B} x {W}x{H}, finest-level iteration launch (last of the call)", "blocks": int(nblk),
           "diagnostic_launch_us": round(launch_us, 1), "chained_tiles_counted": count, "cycles_per_tile_per_wave": round(tot / max(count, 1), 1), "sections": []}
    print(f"{count} chained (non-first) tiles of {nblk} blocks; mean wave cycles per tile {tot / max(count, 1):.0f}")
    for i in ORDER:
        c = sums[i] / max(count, 1)
        f = first_tile[i] / max(n_first, 1)
        print(f"  [{i:2d}] {c:8.0f} cyc {100 * sums[i] / tot:5.1f} %   (first tile of a segment: {f:7.0f})  {SECTION[i]}")
        out["sections"].append({"stamp": i, "what": SECTION[i], "mean_cycles": round(c, 1),
                                "share": round(sums[i] / tot, 4), "first_tile_mean_cycles": round(f, 1)})
    waits = sum(sums[i] for i in (7, 9, 11, 14)) / tot
    print(f"barrier sections together: {100 * waits:.1f} % of a wave's tile time")
    out["barrier_share"] = round(waits, 4)
    # one block's timeline (cycles from the block's first stamp), wave by wave
    b = int(np.argmax(ntile[:, 0] >= 8)) if (ntile[:, 0] >= 8).any() else int(np.argmax(ntile[:, 0]))
    t0 = int(raw[b, :, 0, 0].min())
    tl = []
    for w in range(4):
        row = []
        for t in range(min(ntile[b, 0], 3)):
            row.append({str(i): int((int(raw[b, w, t, i]) - t0) & 0xffffffff) for i in ORDER})
        tl.append(row)
    out["sample_block"] = {"block": b, "tiles": int(ntile[b, 0]), "stamps_by_wave_first_3_tiles": tl}
    for w in range(4):
        print(f"  block {b} wave {w} tile 1:", " ".join(f"{i}:{tl[w][1][str(i)]}" for i in ORDER) if len(tl[w]) > 1 else tl[w])
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1))
    plan.close()


if __name__ == "__main__":
    main()

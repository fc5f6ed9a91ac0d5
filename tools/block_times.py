#!/usr/bin/env python3
"""Block lifetimes of the finest-level fused LK iteration on the chip-wide 100 MHz clock (development tool).

Needs the diagnostic build with the in-tile stamps compiled out (never the shipped library):
    make -C optical-flow-fpga_amd/csrc blocktimes        # -> tools/liboflk_bt.so
    OFLK_LIB=tools/liboflk_bt.so python3 tools/block_times.py [out.json]

Every block of the last finest-level launch records s_memrealtime at kernel entry, at the start and the end
of its tile loop and at exit, and the compute unit it ran on (HW_ID, XCC_ID).  From those: how much of the
launch a compute unit's four block slots are occupied, what a block spends outside its tiles, how long a
slot stays empty between two blocks, and how ragged the end of the launch is.
"""
import collections
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))
TICK_US = 0.01   # s_memrealtime: 100 MHz


def main():
    import torch

    import _oflk
    from oflk_synth import synth_pair

    L = _oflk.lib()
    L.oflk_debug_block_times.restype = ctypes.c_long
    L.oflk_debug_block_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
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
    nblk = L.oflk_debug_block_times(plan._h, None, 0)
    raw = np.zeros((nblk, 8), np.uint32)
    L.oflk_debug_block_times(plan._h, raw.ctypes.data_as(ctypes.c_void_p), nblk)
    raw = raw[raw[:, 7] == 1]
    t = raw[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * TICK_US                                   # us since the first block's entry
    hw, xcc, ntile = raw[:, 4], raw[:, 5] & 0xF, raw[:, 6].astype(int)
    cu_key = (xcc.astype(np.int64) << 16) | (((hw >> 13) & 7).astype(np.int64) << 8) | (((hw >> 12) & 1).astype(np.int64) << 4) | ((hw >> 8) & 15)
    span = t[:, 3].max()
    res = {
        "launch_us_hip_events": round(launch_us, 1), "span_us_first_entry_to_last_exit": round(float(span), 1),
        "blocks": int(len(raw)), "compute_units_seen": int(len(set(cu_key.tolist()))), "tiles": int(ntile.sum()),
        "per_block_us": {"prologue": round(float((t[:, 1] - t[:, 0]).mean()), 2), "tile_loop": round(float((t[:, 2] - t[:, 1]).mean()), 2),
                         "epilogue": round(float((t[:, 3] - t[:, 2]).mean()), 2),
                         "per_tile": round(float(((t[:, 2] - t[:, 1]) / np.maximum(ntile, 1)).mean()), 2)},
    }
    # occupancy of a compute unit's block slots over the span, empty-slot gaps, and the tail
    occ, gaps, last_exit, first_entry = [], [], [], []
    for key in set(cu_key.tolist()):
        m = cu_key == key
        ent, ext = np.sort(t[m, 0]), np.sort(t[m, 3])
        occ.append(float((t[m, 3] - t[m, 0]).sum() / (4.0 * span)))
        last_exit.append(float(ext[-1]))
        first_entry.append(float(ent[0]))
        # the i-th exit frees a slot that the (4 + i)-th entry fills (4 slots per CU)
        n = len(ent)
        for i in range(n - 4):
            gaps.append(float(ent[4 + i] - ext[i]))
    res["slot_occupancy_mean"] = round(float(np.mean(occ)), 3)
    res["slot_occupancy_min_max"] = [round(float(np.min(occ)), 3), round(float(np.max(occ)), 3)]
    res["slot_refill_gap_us"] = {"mean": round(float(np.mean(gaps)), 2), "p50": round(float(np.median(gaps)), 2),
                                 "p90": round(float(np.percentile(gaps, 90)), 2)}
    res["first_entry_us"] = {"mean": round(float(np.mean(first_entry)), 2), "max": round(float(np.max(first_entry)), 2)}
    res["cu_last_exit_us"] = {"min": round(float(np.min(last_exit)), 1), "mean": round(float(np.mean(last_exit)), 1), "max": round(float(np.max(last_exit)), 1)}
    # resident blocks over time, chip-wide (in 5 us bins)
    bins = np.arange(0.0, span + 5.0, 5.0)
    resident = np.zeros(len(bins))
    for a, b in zip(t[:, 0], t[:, 3]):
        i0, i1 = int(a // 5), int(b // 5)
        resident[i0:i1 + 1] += 1
    res["resident_blocks_5us_bins"] = [int(x) for x in resident]
    by_xcc = collections.Counter(xcc.tolist())
    res["blocks_per_xcc"] = {str(k): int(v) for k, v in sorted(by_xcc.items())}
    res["xcc_last_exit_us"] = {str(k): round(float(t[xcc == k, 3].max()), 1) for k in sorted(by_xcc)}
    print(json.dumps(res))
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(res, indent=1))
        np.save(Path(sys.argv[1]).with_suffix(".raw.npy"), raw)


if __name__ == "__main__":
    main()

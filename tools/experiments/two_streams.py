#!/usr/bin/env python3
"""Experiment: does running a batch as two half-batches on two streams (independent frame pairs, so no
dependency between the halves) hide the ramp / tail of the 15 launches of a pyramidal call?
Usage on the GPU box: python3 tools/experiments/two_streams.py [arith: 0 exact | 2 tolerant] [pairs]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "optical-flow-fpga_amd" / "python"))


def main():
    import torch

    import _oflk
    from oflk_synth import synth_pair

    dev = torch.device("cuda", 0)
    arith = int(sys.argv[1]) if len(sys.argv) > 1 else 0     # 0 exact, 2 tolerant
    B, H, W = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 1080, 1920
    host = [synth_pair(H, W, i) for i in range(4)]
    prev = torch.stack([torch.from_numpy(host[b % 4][0]) for b in range(B)]).to(dev)
    curr = torch.stack([torch.from_numpy(host[b % 4][1]) for b in range(B)]).to(dev)
    u, v = torch.empty_like(prev), torch.empty_like(prev)
    N = H * W * 4
    for nsplit in (1, 2, 4, 1, 2, 4):
        Bs = B // nsplit
        plans = [_oflk.Plan(0, Bs, H, W, 3, 5, 3) for _ in range(nsplit)]
        for pl in plans:
            pl.set_arithmetic(arith)
        streams = [torch.cuda.Stream() for _ in range(nsplit)]

        def step():
            for i, (pl, st) in enumerate(zip(plans, streams)):
                o = i * Bs * N
                pl.pyramidal(prev.data_ptr() + o, curr.data_ptr() + o, u.data_ptr() + o, v.data_ptr() + o, st.cuda_stream)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"arith {arith}: {nsplit} stream(s) x {Bs} pairs: {dt * 1e6:.0f} us per {B}-pair step, {B * H * W / dt / 1e6:.0f} Mpix/s", flush=True)
        for pl in plans:
            pl.close()


if __name__ == "__main__":
    main()

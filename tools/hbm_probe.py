#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this box (development tool): torch elementwise ops over 8K planes.
Gives the practical ceiling to read the roofline fractions against (the 8 TB/s peak is never reached by a 1:1 read/write stream)."""
import time

import torch


def main():
    dev = torch.device("cuda", 0)
    n = 4320 * 7680
    a = torch.rand(n, device=dev)
    b = torch.rand(n, device=dev)
    c = torch.empty_like(a)
    for name, fn, nbytes in (("c = a + b (2 reads, 1 write)", lambda: torch.add(a, b, out=c), 12 * n),
                             ("c = a (1 read, 1 write)", lambda: c.copy_(a), 8 * n),
                             ("a.sum() (1 read)", lambda: a.sum(), 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"{name}: {dt * 1e6:.1f} us, {nbytes / dt / 1e12:.2f} TB/s")


if __name__ == "__main__":
    main()

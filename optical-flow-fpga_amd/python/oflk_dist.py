"""Multi-GPU plumbing: one process per GPU, frame pairs sharded across ranks.

Frame pairs are independent units (reference lucas_kanade_pyramidal.py:141-228
touches only its two inputs), so the data path needs NO collective: every rank
runs the same plan on its own pairs.  torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in CPU tests) is used only for the barrier that
brackets a timed region, the MAX over ranks of the elapsed time, and an optional
gather of small per-rank summaries to rank 0.
"""
from __future__ import annotations

import os
from typing import Any, List, Optional, Tuple


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` units owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(total), world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class Group:
    """Thin wrapper so single-process runs need no process group."""

    def __init__(self, backend: Optional[str] = None, device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist

            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj: Any) -> Optional[List[Any]]:
        """Small Python summaries to rank 0 (None elsewhere)."""
        if self.dist is None:
            return [obj]
        out: Optional[List[Any]] = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def close(self) -> None:
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.dist = None

"""Multi-GPU plumbing: one process per GPU, frame pairs sharded across ranks.

Frame pairs are independent units (reference lucas_kanade_pyramidal.py:141-228
touches only its two inputs), so the data path needs NO collective: every rank
runs the same plan on its own pairs.  torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in CPU tests) is used only for the barrier that
brackets a timed region, the MAX over ranks of the elapsed time, a gather of small
per-rank summaries to rank 0 and -- optional, after the timed region, reported on its
own -- the gather of the flow shards themselves (gather_flows).
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple


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

    def __init__(self, backend: Optional[str] = None, device=None, always: bool = False):
        """always: create the process group even for one rank (a one-GPU box can then run the RCCL barrier / MAX / SUM
        this module uses; tests/test_gpu_round3.py)."""
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1 or always:
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


# ---------------------------------------------------------------------------
# bench.py's job layout and timed region (here so that the CPU tests drive the same code)
# ---------------------------------------------------------------------------
CONFIGS: Dict[str, Dict[str, Any]] = {
    # BASELINE.json configs[2], the config the metric is quoted on: per-GPU work fixed (weak scaling)
    # 128 pairs per GPU and step: large batches amortise the ragged end of every launch (32 pairs: 23.3, 64: 23.6,
    # 128: 24.0 Gpix/s on one box); 18 GB of the 288 GB
    "1080p": {"height": 1080, "width": 1920, "pairs_per_gpu": 128, "total_pairs": None, "scaling": "weak",
              "label": "BASELINE configs[2]: 1920x1080 frame pairs, batches per GPU"},
    # BASELINE.json configs[3]: ONE job of 64 pairs of 3840x2160 cut over the ranks (strong scaling)
    "4k64": {"height": 2160, "width": 3840, "pairs_per_gpu": None, "total_pairs": 64, "scaling": "strong",
             "label": "BASELINE configs[3]: 64 frame pairs of 3840x2160 sharded over the GPUs"},
}


@dataclass
class JobLayout:
    config: str
    height: int
    width: int
    pairs_local: int      # pairs this rank processes per step
    pair_begin: int       # index of its first pair in the job
    pairs_total: int      # pairs of the whole job per step
    scaling: str          # "weak" | "strong"
    label: str


def job_layout(config: str, rank: int, world: int, pairs_per_gpu: Optional[int] = None,
               height: Optional[int] = None, width: Optional[int] = None) -> JobLayout:
    """Which frame pairs a rank owns.  Pairs are independent units: ranks never exchange data."""
    if config not in CONFIGS:
        raise ValueError(f"unknown config {config!r} (have {sorted(CONFIGS)})")
    c = CONFIGS[config]
    H, W = int(height or c["height"]), int(width or c["width"])
    if c["scaling"] == "weak":
        n = int(pairs_per_gpu or c["pairs_per_gpu"])
        return JobLayout(config, H, W, n, rank * n, world * n, "weak", c["label"])
    total = int(pairs_per_gpu * world) if pairs_per_gpu else int(c["total_pairs"])
    b0, b1 = shard_range(total, rank, world)
    return JobLayout(config, H, W, b1 - b0, b0, total, "strong", c["label"])


def run_timed(group: "Group", step: Callable[[], None], device_sync: Callable[[], None], steps: int, warmup: int,
              before_timed: Optional[Callable[[], None]] = None) -> float:
    """bench.py's timed region: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by
    device-sync + barrier + device-sync on both sides; returns the MAX over ranks of the elapsed seconds."""

    def fence() -> None:
        device_sync()
        group.barrier()
        device_sync()

    for _ in range(warmup):
        step()
    fence()
    if before_timed is not None:
        before_timed()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    elapsed = time.perf_counter() - t0
    fence()
    return group.max_over_ranks(elapsed)


def job_throughput(group: "Group", layout: JobLayout, steps: int, elapsed_max: float, local_sums: Dict[str, float]
                   ) -> Dict[str, Any]:
    """Whole-job figures on every rank: pixels of ALL ranks / MAX-rank time, and SUM-reduced result totals."""
    pairs_all = int(round(group.sum_over_ranks(float(layout.pairs_local))))
    sums = {k: group.sum_over_ranks(float(v)) for k, v in sorted(local_sums.items())}
    pix = float(pairs_all) * layout.height * layout.width * steps
    return {"pairs_per_step": pairs_all, "Mpix_per_s": pix / elapsed_max / 1e6, "sums": sums}


def gather_flows(group: "Group", u, v, layouts: List[JobLayout], device_sync: Callable[[], None], dst: int = 0
                 ) -> Dict[str, Any]:
    """The flow shards of all ranks to rank `dst`, as ONE collective on the group's backend (RCCL gather on device tensors,
    gloo on host tensors in the tests): never part of the compute time -- bench.py runs it after the timed region and reports
    it on its own.  u, v: this rank's [pairs_local, H, W] float32 tensors (on the group's device); layouts: the JobLayout of
    every rank (job_layout is a pure function of rank and world, so every rank can list them).
    Returns {"gather_ms", "gather_GBs", "bytes_received", "pairs"} and, on `dst`, "u" / "v": [pairs_total, H, W] tensors in
    job order.  A gather wants equal pieces: shards are padded to the largest one (they differ by at most one pair)."""
    import torch

    me = layouts[group.rank]
    n_max = max(l.pairs_local for l in layouts)
    H, W = me.height, me.width
    send = torch.zeros((2, n_max, H, W), dtype=torch.float32, device=u.device)
    send[0, :me.pairs_local].copy_(u)
    send[1, :me.pairs_local].copy_(v)
    pieces = [torch.empty_like(send) for _ in layouts] if group.rank == dst else None
    device_sync()
    group.barrier()
    device_sync()
    t0 = time.perf_counter()
    if group.dist is not None:
        group.dist.gather(send, pieces, dst=dst)
    else:
        pieces[0].copy_(send)
    device_sync()
    elapsed = group.max_over_ranks(time.perf_counter() - t0)
    received = sum(2 * l.pairs_local * H * W * 4 for r, l in enumerate(layouts) if r != dst)   # useful bytes that crossed ranks
    out: Dict[str, Any] = {"gather_ms": 1e3 * elapsed, "gather_GBs": received / elapsed / 1e9 if elapsed > 0 else 0.0,
                           "bytes_received": received, "pairs": sum(l.pairs_local for l in layouts)}
    if group.rank == dst:
        out["u"] = torch.cat([pieces[r][0, :l.pairs_local] for r, l in enumerate(layouts)])
        out["v"] = torch.cat([pieces[r][1, :l.pairs_local] for r, l in enumerate(layouts)])
    return out

"""Multi-GPU layout of the path: per-GPU stream pools, no data-path collective (SURVEY §8 e).

Streams are independent (one ``(h, c)`` + one state machine each, never reading another stream's
data: /root/reference/websocket_service/server/vad_websocket_server.py:277), so N GPUs are N
independent engines.  ``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests) is used for
exactly two control-plane operations around a timed region: a barrier and a MAX-reduction of
the elapsed time.  Nothing here moves audio, state or probabilities between ranks.
"""

from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Callable, Optional, Tuple


@dataclass(frozen=True)
class RankInfo:
    rank: int
    local_rank: int
    world: int

    @classmethod
    def from_env(cls) -> "RankInfo":
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                   int(os.environ.get("WORLD_SIZE", "1")))


def stream_shard(total_streams: int, world: int, rank: int) -> Tuple[int, int]:
    """Static contiguous partition: rank r owns global stream ids [lo, hi).  ``gpu = id // per_gpu``
    with the remainder spread over the first ranks (SURVEY §8 e, "Partitioning")."""
    if world < 1 or not (0 <= rank < world) or total_streams < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(stream_id: int, total_streams: int, world: int) -> int:
    """Inverse of :func:`stream_shard`: which rank serves a global stream id."""
    base, extra = divmod(total_streams, world)
    edge = extra * (base + 1)
    if stream_id < edge:
        return stream_id // (base + 1)
    return extra + (stream_id - edge) // max(base, 1)


def init_process_group(info: RankInfo, backend: str, device=None):
    """None for a single process; otherwise the initialised ``torch.distributed`` module."""
    if info.world == 1:
        return None
    import torch.distributed as dist
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=info.rank, world_size=info.world, **kw)
    return dist


def timed_region(dist, run: Callable[[], None], sync: Callable[[], None], device: Optional[str] = None) -> float:
    """barrier + sync | run() | sync + barrier, then MAX of the wall time over ranks (seconds)."""
    import torch
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_rate(units_per_rank_per_step: int, steps: int, world: int, elapsed_max: float) -> float:
    """Whole-job throughput under weak scaling: every rank did the same number of units."""
    return world * units_per_rank_per_step * steps / elapsed_max

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


class ControlPlane:
    """The only three things ranks ever say to each other: barrier, MAX of a number, gather of small records.

    The default process group is ALWAYS gloo (it needs nothing but the rendezvous).  With ``prefer="nccl"`` an RCCL group over
    the ranks' GPUs is created on top and probed with one all-reduce; every rank then reports over gloo whether its probe
    worked, and only if ALL did is RCCL used for the barrier and the MAX.  Any failure (no GPU, two ranks on one device, an
    RCCL that cannot initialise on this node) leaves the job on gloo, in the same process, with the reason kept in
    ``fallback_reason`` - the measurement does not depend on the choice, there is no data-path collective either way."""

    def __init__(self, info: RankInfo, prefer: str = "gloo", device=None, probe_timeout_s: float = 60.0) -> None:
        self.info = info
        self.backend: Optional[str] = None
        self.fallback_reason: Optional[str] = None
        self._dist = None
        self._group = None
        self._device = None
        if info.world == 1:
            return
        import datetime
        import torch
        import torch.distributed as dist
        # gloo advertises the address its hostname resolves to; a container whose hostname does not resolve would fail right here,
        # so a single-node job (rendezvous on the loopback address) is then pinned to the loopback interface
        if "GLOO_SOCKET_IFNAME" not in os.environ and os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
            import socket
            try:
                socket.gethostbyname(socket.gethostname())
            except OSError:
                os.environ["GLOO_SOCKET_IFNAME"] = "lo"
        dist.init_process_group("gloo", rank=info.rank, world_size=info.world)
        self._dist = dist
        self.backend = "gloo"
        if prefer != "nccl":
            return
        ok, why, group = 1, "", None
        try:
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")       # a stuck probe raises after the timeout instead of aborting the process
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=probe_timeout_s))
            t = torch.ones(1, device=device)
            dist.all_reduce(t, group=group)
            torch.cuda.synchronize()
            if int(t.item()) != info.world:
                raise RuntimeError(f"probe all-reduce returned {t.item()} for world size {info.world}")
        except Exception as e:      # noqa: BLE001 - whatever RCCL throws, the job carries on over gloo
            ok, why = 0, f"{type(e).__name__}: {e}".splitlines()[0][:300]
        flags = [None] * info.world
        dist.all_gather_object(flags, (ok, why))
        if all(f[0] for f in flags):
            self._group, self._device, self.backend = group, device, "nccl"
        else:
            bad = [(r, f[1]) for r, f in enumerate(flags) if not f[0]]
            self.fallback_reason = f"rank {bad[0][0]}: {bad[0][1]}" + (f" (+{len(bad) - 1} more ranks)" if len(bad) > 1 else "")

    @property
    def active(self) -> bool:
        return self._dist is not None

    def barrier(self) -> None:
        if self._dist is None:
            return
        if self._group is not None:
            self._dist.barrier(group=self._group, device_ids=[self._device.index])
        else:
            self._dist.barrier()

    def max(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self._device if self._group is not None else "cpu")
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self._group)
        return float(t.item())

    def gather(self, record) -> list:
        """every rank's record on every rank, in rank order (gloo: small python objects)"""
        if self._dist is None:
            return [record]
        out = [None] * self.info.world
        self._dist.all_gather_object(out, record)
        return out

    def close(self) -> None:
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None


def timed_region(dist, run: Callable[[], None], sync: Callable[[], None], device: Optional[str] = None) -> float:
    """barrier + sync | run() | sync + barrier, then MAX of the wall time over ranks (seconds).  ``dist``: None, an initialised
    ``torch.distributed`` module, or a :class:`ControlPlane`."""
    if isinstance(dist, ControlPlane):
        return timed_region_detail(dist, run, sync)[0]
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


def timed_region_detail(cp: ControlPlane, run: Callable[[], None], sync: Callable[[], None]) -> Tuple[float, float]:
    """-> (MAX over ranks, this rank's own) wall time of: sync, barrier, sync | run() | sync, barrier."""
    sync()
    cp.barrier()
    sync()
    t0 = time.perf_counter()
    run()
    sync()
    own = time.perf_counter() - t0
    cp.barrier()
    return cp.max(own), own


def duplicate_devices(records: list) -> list:
    """records = [(host, device identity), ...] in rank order -> [(rank_a, rank_b, identity)] for every pair of ranks that
    would step the same GPU (a wrong LOCAL_RANK -> device map halves the hardware under the reported n_gpus)."""
    seen, dup = {}, []
    for r, key in enumerate(records):
        key = tuple(key)
        if key in seen:
            dup.append((seen[key], r, key))
        else:
            seen[key] = r
    return dup


def aggregate_rate(units_per_rank_per_step: int, steps: int, world: int, elapsed_max: float) -> float:
    """Whole-job throughput under weak scaling: every rank did the same number of units."""
    return world * units_per_rank_per_step * steps / elapsed_max

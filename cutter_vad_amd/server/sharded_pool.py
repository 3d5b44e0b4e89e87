"""Sessions of one process over SEVERAL GPUs: one stream pool - one engine, one ticker thread - per device.

SURVEY §8 e: independent audio streams are the data-parallel axis, "a host thread (or process) per GPU drives its pool", no
collective.  The reference's anchor is one ``VADWrapper`` per websocket client
(/root/reference/websocket_service/server/vad_websocket_server.py:277): a client's stream never reads another's, so where it
lives is a placement decision only.  ``ShardedStreamPool`` places a new session on the least-loaded shard and hands back the
same :class:`PooledSession` a single pool would; every shard ticks on its own thread, so the devices step concurrently and a
slow device does not hold the others' events back.

``migrate(session, shard)`` moves a live session to another shard - in the middle of an utterance if need be: the frames it
still has queued are stepped where it is (order is kept), then its recurrent state + state machine (``vad_stream_save``) and
its segment audio (``vad_tick_segment_save``: pre-roll, open segment, finished one not yet taken) are restored on a fresh slot
of the target engine; the session object, its callbacks and its counters carry over.  Results are bit-identical to a session
that never moved (tests/test_gpu_server.py).  ``rebalance()`` uses it to even the shards out after clients have left.
"""

from __future__ import annotations

import threading
import time
from typing import List, Optional, Sequence

from ..core.config import SileroModelVersion, VADConfig
from ..core.exceptions import AudioProcessingError
from ..pool import EnginePool
from . import shared_pool
from .shared_pool import PooledSession, SharedStreamPool, conduct_ticks


class ShardedStreamPool:
    """``devices``: HIP device ordinals, one shard each (the same ordinal twice = two engines on one GPU, which is how the
    single-GPU test box exercises it).  ``shards``: ready-made pools instead (tests on the CPU doubles)."""

    def __init__(self, devices: Optional[Sequence[int]] = None, model_version: SileroModelVersion = SileroModelVersion.V5,
                 max_streams: Optional[int] = None, tick_interval: float = 0.010, sample_rate: int = 16000,
                 convert_rates: bool = False, shards: Optional[Sequence[SharedStreamPool]] = None) -> None:
        if shards is None:
            if not devices:
                raise AudioProcessingError("ShardedStreamPool needs at least one device")
            # every shard gets its own engine registry: two shards never share an engine, even on one device
            shards = [SharedStreamPool(model_version=model_version, device_id=int(d), max_streams=max_streams, pool=EnginePool(),
                                       tick_interval=tick_interval, sample_rate=sample_rate, convert_rates=convert_rates)
                      for d in devices]
        self.shards: List[SharedStreamPool] = list(shards)
        if not self.shards:
            raise AudioProcessingError("ShardedStreamPool needs at least one shard")
        self.devices = list(devices) if devices is not None else list(range(len(self.shards)))
        self.frame = self.shards[0].frame
        self.convert_rates = self.shards[0].convert_rates
        self.tick_interval = tick_interval
        # a process that serves several GPUs builds tens of MB of voice_end payloads per tick: the allocator keeps what a tick frees
        # (csrc/wirebox.c, keep_heap - otherwise every tick's payloads are fresh pages, faulted in by all shards' threads at once)
        if shared_pool._wirebox is not None and len(self.shards) > 1:
            shared_pool._wirebox.keep_heap(1 << 30, 256 << 20)
        self._place = threading.Lock()             # placement and migration: one at a time
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()
        self.migrations = 0

    # ------------------------------------------------------------------ sessions (the SharedStreamPool surface)
    def shard_of(self, s: PooledSession) -> int:
        return self.shards.index(s.pool)

    def open_session(self, config: Optional[VADConfig] = None, shard: Optional[int] = None) -> PooledSession:
        with self._place:
            k = min(range(len(self.shards)), key=lambda i: self.shards[i].session_count) if shard is None else int(shard)
            return self.shards[k].open_session(config)

    def close_session(self, s: PooledSession) -> None:
        with self._place:                          # not while the session is between two shards: `s.pool` is stable in here
            s.pool.close_session(s)

    def reconfigure(self, s: PooledSession, config: VADConfig) -> None:
        with self._place:
            s.pool.reconfigure(s, config)

    @property
    def session_count(self) -> int:
        return sum(p.session_count for p in self.shards)

    @property
    def backlog(self) -> int:
        return sum(p.backlog for p in self.shards)

    # ------------------------------------------------------------------ moving a live session
    def migrate(self, s: PooledSession, shard: int) -> None:
        dst = self.shards[int(shard)]
        with self._place:
            src = s.pool
            if s.closed:
                raise AudioProcessingError("session is closed")
            if dst is src:
                return
            # 1. nothing new reaches an engine for the session while it moves: frames submitted meanwhile are held on the session
            #    (producers do not wait - one of them may be an event loop) and replayed on the pool it lands on; what is already
            #    queued is stepped where the session lives (its events are delivered as usual, in order)
            first, second = sorted((src, dst), key=id)
            with src._lock:
                src._unbind_push(s)
                src._flush_inbox()
                s.moving = True
            try:
                for _ in range(300):                        # at most 256 frames wait for one stream, one is stepped per tick
                    if src.engine.tick_pending(s.slot) == 0:
                        break
                    src.tick()
                else:
                    raise AudioProcessingError("session's queued frames could not be stepped on its current engine; not moved")
                with first._tick_lock, second._tick_lock:
                    if src.engine.tick_pending(s.slot) > 0:      # a frame slipped in between the last tick and the lock
                        raise AudioProcessingError("session kept receiving frames while it was being moved")
                    state = src.engine.save_stream(s.slot)
                    audio = src.engine.save_segment(s.slot)
                    new_slot = int(dst.engine.open_stream())
                    try:
                        dst.engine.tick_cancel(new_slot)
                        dst.engine.restore_stream(new_slot, state)            # (h, c), thresholds, counters, history
                        dst.engine.restore_segment(new_slot, audio)
                    except Exception:
                        dst.engine.close_stream(new_slot)
                        raise
                    old_slot = s.slot
                    with src._lock, dst._lock:
                        dst._grow(new_slot + 1)
                        for name in ("_thr", "_active", "_cont", "_contp", "_gate", "_lastp", "_done"):
                            getattr(dst, name)[new_slot] = getattr(src, name)[old_slot]
                        dst._cont_cb[new_slot], src._cont_cb[old_slot] = src._cont_cb[old_slot], None
                        dst._wav_rate[new_slot], src._wav_rate[old_slot] = src._wav_rate[old_slot], 0
                        src._sessions.pop(old_slot, None)
                        src._by_slot[old_slot] = None
                        s._home = (dst, new_slot)      # readers of the pair (frames_done, active) never see a mixed one
                        s.pool, s.slot = dst, new_slot
                        dst._sessions[new_slot] = s
                        dst._by_slot[new_slot] = s
                    src.engine.tick_cancel(old_slot)
                    src.engine.close_stream(old_slot)
                    self.migrations += 1
            finally:
                with s.pool._lock:                     # on the pool it ended up on
                    s.pool._replay_held(s)             # what arrived meanwhile, in order, before anything newer can get in
                    s.pool._bind_push(s)
                    s.moving = False

    def rebalance(self, tolerance: int = 1) -> int:
        """Move sessions from the fullest to the emptiest shard until they differ by at most ``tolerance``; -> moves made."""
        moved = 0
        while True:
            counts = [p.session_count for p in self.shards]
            hi, lo = max(range(len(counts)), key=counts.__getitem__), min(range(len(counts)), key=counts.__getitem__)
            if counts[hi] - counts[lo] <= max(1, tolerance):
                return moved
            victim = next((x for x in self.shards[hi]._sessions.values() if not x.closed), None)
            if victim is None:
                return moved
            self.migrate(victim, lo)
            moved += 1

    # ------------------------------------------------------------------ ticking
    def tick(self) -> int:
        """One tick on every shard, the devices side by side, conducted by this thread (``shared_pool.conduct_ticks``: the
        engines' work on one C thread per shard behind a single release of the interpreter lock, then the events pool by pool);
        -> frames processed."""
        return conduct_ticks(self.shards)

    def drain(self, max_ticks: int = 1 << 30) -> int:
        total = 0
        for _ in range(max_ticks):
            n = self.tick()
            if n == 0:
                break
            total += n
        return total

    def start(self) -> None:
        """ONE ticker thread for all shards: every ``tick_interval`` it conducts a tick of every pool (``tick``).  (A ticker
        thread per pool, as round 3 had it, made the threads queue for the interpreter lock: DESIGN.md §4.4.)"""
        if self._thread is not None:
            return
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                t0 = time.perf_counter()
                try:
                    self.tick()
                except Exception:                   # the ticker outlives any single bad tick
                    pass
                if self.backlog:                    # a client is ahead of real time: catch up, do not sleep
                    continue
                self._stop.wait(max(0.0, self.tick_interval - (time.perf_counter() - t0)))

        self._thread = threading.Thread(target=loop, name="vad-shards-ticker", daemon=True)
        self._thread.start()

    def stop(self) -> None:
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def close(self) -> None:
        self.stop()
        for p in self.shards:
            p.close()

    def stats(self) -> dict:
        per = [p.stats() for p in self.shards]
        frames, launches = sum(x["frames"] for x in per), sum(x["launches"] for x in per)
        return {"sessions": self.session_count, "ticks": sum(x["ticks"] for x in per), "frames": frames, "launches": launches,
                "frames_per_launch": frames / launches if launches else 0.0, "migrations": self.migrations,
                "shards": [dict(x, device=d) for x, d in zip(per, self.devices)]}

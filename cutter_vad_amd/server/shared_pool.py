"""All client sessions of a process on ONE stream pool: the multi-stream caller the engine is built for.

The reference gives every websocket client its own ``VADWrapper`` (own ORT session, own lock) and runs the
model inline in the socket's receive loop, one frame at a time
(/root/reference/websocket_service/server/vad_websocket_server.py:277, :369).  Here a session is one slot of
the shared engine; ``submit()`` only queues a frame, and a ticker advances EVERY session that has a frame
pending with one ``vad_step_events`` launch (gate + model + state machine on the device for the whole
batch), then fans the events out to the sessions' callbacks.

Per-session semantics are those of the reference's ``VADWrapper`` with ``buffer_size`` = the client's frame
length (``vad_websocket_server.py:253-266``): frames shorter than 512 samples are zero-padded on the right
(core/silero_model.py:464-468), longer ones truncated; the denoise gate applies to the model input (in the
kernel) and to the audio kept for the segment (utils/audio.py:104-121); START / CONTINUE / END follow
``_process_voice_state`` (core/silero_model.py:790-949) — the decisions come from the device state machine,
the host only keeps the audio: pre-roll = the above-threshold frames before START (:838-869), CONTINUE
carries the frame's float32 bytes (:891-895), END carries the WAV of the segment (:925-949).
A session's frames are processed in submission order, at most one per tick.

The tick itself - queueing, padding, grouping by (wire format, gate), staging in page-locked memory, the launches and the
compaction of the results - runs in C behind ``vad_tick_push`` / ``vad_tick_run`` (include/vad_engine.h): ``submit*`` writes
a frame straight into the coming tick's staging row, ``tick`` gets back arrays over the stepped streams.  Python works on
those arrays at once; the segments' audio (pre-roll, open segment) is kept by the engine's tick as well
(``vad_tick_enable_segments``), so a session is touched individually only on START, on END, or while it talks if it asked
for ``voice_continue`` payloads.

``convert_rates=True`` (opt-in; Silero V5 pools at 16 kHz): a session whose config names 8 / 24 / 48 kHz with
``auto_convert_sample_rate`` sends chunks of ``512 * sample_rate / 16000`` samples (32 ms) at its own rate; the tick resamples
them on the GPU (``AudioUtils.resample_audio`` = scipy's Fourier method, utils/audio.py:46-49, as one operator per rate) and
steps them in the same launch (``vad_tick_push_rate``).  The reference declares that conversion but leaves the hook empty
(vad_wrapper.py:621-624) and its V5 graph fails on such frames; with the flag off this pool fails the same way.  Segments
and ``voice_continue`` payloads carry the audio at the rate it arrived in.
"""

from __future__ import annotations

import bisect
import ctypes as C
import os
import threading
import time
from collections import deque
from typing import Callable, Deque, Dict, List, Optional

import numpy as np

from .. import _ffi, weights_io
from ..core.config import SileroModelVersion, VADConfig
from ..core.exceptions import AudioProcessingError, CallbackError
from ..core.silero_model import SileroVADModel
from ..engine import TICK_RATES
from ..pool import EnginePool, default_pool, resolve_model_path
from ..utils.audio import AudioUtils
from ..utils.wav_writer import WAVWriter

def _load_wirebox():
    """The C inbox for wire frames (csrc/wirebox.c), built by ``__graft_entry__.build()`` / ``python -m cutter_vad_amd._build``.
    Optional: without it frames queue in a Python dict - same results, ~10 x the per-frame cost.  Nothing is compiled here."""
    if os.environ.get("VAD_POOL_WIREBOX", "1") == "0":
        return None
    try:
        from .. import _wirebox
        return _wirebox
    except ImportError:
        return None


_wirebox = _load_wirebox()
_RETRY = object()       # submit's answer while a session is between two pools
FRAME = 512      # the model's frame at 16 kHz; a pool's own frame length is ``SharedStreamPool.frame`` (256 on V5's 8 kHz sub-model)


class PooledSession:
    """One client stream: a slot of the shared engine + the host half of its voice segments."""

    __slots__ = ("pool", "slot", "_home", "config", "long_frames", "on_start", "on_end", "on_continue", "on_error", "closed",
                 "wav_writer", "user", "rate", "moving", "gate", "_push", "_queued", "lost", "_held", "submit_pcm16")

    def __init__(self, pool: "SharedStreamPool", slot: int, config: VADConfig) -> None:
        self.pool = pool
        self.slot = slot
        self._home = (pool, slot)                  # ONE attribute for readers that need the pair (migrate stores it in one go)
        self.lost = 0                              # frames taken from this session and never stepped (failed tick, refused push)
        self._held: Deque = deque()                # frames that arrived while the session was changing engines (migrate replays them)
        self.config = config
        self.long_frames: Deque = deque()          # frames longer than the model's frame, whole: segments keep all of a frame
        self.on_start: Optional[Callable[[], None]] = None
        self.on_end: Optional[Callable[[bytes], None]] = None
        self.on_continue: Optional[Callable[[bytes], None]] = None
        self.on_error: Optional[Callable[[Exception], None]] = None
        self.closed = False
        self.moving = False                        # ShardedStreamPool.migrate: frames wait while the session changes engines
        self.wav_writer = WAVWriter(sample_rate=config.output_wav_sample_rate, bit_depth=config.output_wav_bit_depth,
                                    channels=1)
        self.user = None
        self.rate: Optional[int] = pool._input_rate(config)      # None: frames at the engine's rate; else resampled in the tick
        self.gate = bool(config.enable_denoising)                # plain bool copy of pool._gate[slot] for the per-frame path
        self._push = None                                        # the pool's C inbox bound to this slot (SharedStreamPool._bind_push)
        self._queued = (-1, 0)                                   # (inbox epoch, byte length) of the last frame queued in Python
        # ``submit_pcm16(data)``: one wire frame (little-endian int16 PCM).  An instance attribute: while the session is live and in
        # place it IS the C inbox's pusher bound to this slot (csrc/wirebox.c) - one C call per frame, no interpreter frame in
        # between; a frame the pusher does not take (another length, the session moving, closed, reconfigured) goes to
        # ``_submit_pcm16_general`` from inside that call.  Without the C inbox it is the general path itself.
        self.submit_pcm16 = self._submit_pcm16_general

    # the per-session scalars live in the pool's arrays (indexed by slot) so that a tick can work on all of them at once
    @property
    def active(self) -> bool:
        pool, slot = self._home
        return bool(pool._active[slot])

    @property
    def last_probability(self) -> float:
        pool, slot = self._home
        return float(pool._lastp[slot])

    @property
    def frames_done(self) -> int:
        pool, slot = self._home
        return int(pool._done[slot])

    def set_callbacks(self, voice_start_callback=None, voice_end_callback=None, voice_continue_callback=None,
                      error_callback=None, continue_payload: bool = True) -> None:
        """``continue_payload=False``: the voice_continue callback is only a notification (the reference server's is:
        ``_on_voice_continue`` ignores its ``pcm_data``, vad_websocket_server.py:420-430) - it is called with ``b""`` and the tick does
        not build the frame's float32 bytes for it (≈ 10 µs per talking session and tick)."""
        self.on_start, self.on_end, self.on_continue = voice_start_callback, voice_end_callback, voice_continue_callback
        self.on_error = error_callback
        self.pool._cont[self.slot] = voice_continue_callback is not None
        self.pool._contp[self.slot] = voice_continue_callback is not None and bool(continue_payload)
        self.pool._cont_cb[self.slot] = voice_continue_callback if not continue_payload else None

    def submit(self, frame) -> None:
        # _RETRY: this thread read `self.pool` just before a migration swapped it - the next read sees the new pool.  A session
        # that is still moving does not make its producer wait (it may be an event loop): the frame is held and replayed.
        while self.pool.submit(self, frame) is _RETRY:
            pass

    def _submit_pcm16_general(self, data: bytes) -> None:
        while self.pool.submit_pcm16(self, data) is _RETRY:
            pass

    def is_voice_active(self) -> bool:
        return self.active

    def close(self) -> None:
        self.pool.close_session(self)


class SharedStreamPool:
    def __init__(self, model_version: SileroModelVersion = SileroModelVersion.V5, device_id: Optional[int] = None,
                 max_streams: Optional[int] = None, pool: Optional[EnginePool] = None,
                 tick_interval: float = 0.010, sample_rate: int = 16000, convert_rates: bool = False) -> None:
        vi = 4 if model_version == SileroModelVersion.V4 else 5
        self.frame = weights_io.frame_samples(vi, sample_rate)     # 512; 256 on Silero V5's 8 kHz sub-model
        SileroVADModel._check_rate(sample_rate, model_version, self.frame)
        self._base = VADConfig(model_version=model_version, sample_rate=sample_rate, buffer_size=self.frame)
        self._k8 = weights_io.is_8k_variant(vi, sample_rate)
        self.convert_rates = bool(convert_rates) and vi == 5 and int(sample_rate) == 16000
        self._pool = pool or default_pool()
        self.engine = self._pool.engine_for(resolve_model_path(self._base), model_version, device_id, max_streams,
                                            sample_rate)
        self.engine.tick_enable_segments(True)      # the segments' audio is kept by the engine's tick, not here
        self.tick_interval = tick_interval
        self._lock = threading.Lock()              # sessions / pending queues
        self._tick_lock = threading.Lock()         # one tick at a time
        self._sessions: Dict[int, PooledSession] = {}
        self._by_slot: List[Optional[PooledSession]] = []
        self._cont_cb: List = []                   # slot -> the session's voice_continue callback while it is a NOTIFICATION (else None)
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()
        # per-slot arrays, sized for the whole engine up front: the tick hands them to the engine (vad_tick_run_work updates them
        # with the GIL released), so they must not be re-allocated under it
        self._grow(max(1024, int(getattr(self.engine, "max_streams", 0) or 0)))
        # int16 wire frames no longer than the model's frame - what every websocket client sends - are not pushed one by one:
        # they collect here, per (byte length, gate), and go to the engine as ONE vad_tick_push_status call per key at the
        # start of the next tick
        self._inbox: Dict[tuple, List] = {}
        self._epoch = 0                            # counts the flushes
        # ... and with the _wirebox extension built they do not even take this pool's lock: each session holds a C callable bound
        # to its slot that appends (slot, bytes) to an array, and the tick hands the arrays to vad_tick_push_gather as they are
        self._wire = _wirebox.Inbox(2 * self.frame) if _wirebox is not None else None
        self._wire_entry = getattr(self.engine, "tick_gather_entry", None)    # -> (function address, engine address); None: test doubles
        self._wav_entry = getattr(self.engine, "tick_wav_entry", None) if _wirebox is not None else None
        self.backlog = 0                           # frames already staged for the next tick when the last one returned
        self.ticks = 0
        self.frames = 0
        self.launches = 0

    def _grow(self, n: int) -> None:
        old = getattr(self, "_thr", None)
        k = 0 if old is None else old.size
        if n <= k:
            return
        n = max(n, 2 * k)

        def ext(name, dtype):
            a = np.zeros(n, dtype)
            if k:
                a[:k] = getattr(self, name)
            setattr(self, name, a)
        ext("_thr", np.float64)        # vad_start_probability
        ext("_active", bool)           # inside a segment
        ext("_cont", bool)             # registered a voice_continue callback
        ext("_contp", bool)            # ... that wants the frame's bytes
        ext("_gate", bool)             # enable_denoising
        ext("_lastp", np.float32)
        ext("_done", np.int64)
        ext("_wav_rate", np.int32)     # the session's WAV sample rate if its voice_end payload is 16-bit mono (built by the conductor), else 0
        self._by_slot.extend([None] * (n - len(self._by_slot)))
        self._cont_cb.extend([None] * (n - len(self._cont_cb)))

    def _init_slot(self, slot: int, cfg: VADConfig) -> None:
        self._grow(slot + 1)
        self._thr[slot] = float(cfg.vad_start_probability)
        self._gate[slot] = bool(cfg.enable_denoising)
        self._active[slot] = self._cont[slot] = self._contp[slot] = False
        self._cont_cb[slot] = None
        self._lastp[slot] = 0.0
        self._done[slot] = 0
        self._wav_rate[slot] = (int(cfg.output_wav_sample_rate) if int(cfg.output_wav_bit_depth) == 16 else 0)

    # ------------------------------------------------------------------ sessions
    def open_session(self, config: Optional[VADConfig] = None) -> PooledSession:
        cfg = config or self._base
        if cfg.model_version != self._base.model_version:
            raise AudioProcessingError(f"this pool serves Silero {self._base.model_version.value} streams")
        self._check_session_rate(cfg)
        slot = int(self.engine.open_stream())
        try:
            self.engine.tick_cancel(slot)          # a recycled slot starts with no frames waiting
            self.engine.set_thresholds(slot, cfg.vad_start_probability, cfg.vad_end_probability, cfg.voice_start_ratio,
                                       cfg.voice_end_ratio, cfg.voice_start_frame_count, cfg.voice_end_frame_count)
        except Exception:
            self.engine.close_stream(slot)
            raise
        s = PooledSession(self, slot, cfg)
        with self._lock:
            self._init_slot(slot, cfg)
            self._sessions[slot] = s
            self._by_slot[slot] = s
            self._bind_push(s)
        return s

    def _bind_push(self, s: PooledSession) -> None:
        """Give the session its fast ingest: a callable of the C inbox bound to (slot, gate).  ``_lock`` held."""
        if self._wire is None or s.closed:
            s._push = None
        elif s.rate is None:
            s._push = self._wire.pusher(s.slot, s.gate, 0, 0, s._submit_pcm16_general)
        else:                                       # a client at 8 / 24 / 48 kHz: only its exact 32 ms chunk is taken (int16 bytes)
            s._push = self._wire.pusher(s.slot, s.gate, int(s.rate), 2 * (FRAME * int(s.rate) // 16000), s._submit_pcm16_general)
        s.submit_pcm16 = s._push if s._push is not None else s._submit_pcm16_general

    @staticmethod
    def _unbind_push(s: PooledSession) -> None:
        """From here on the session's frames take the general path (which looks at closed / moving).  Always BEFORE the flush
        that precedes a cancel: nothing can join the C inbox for this slot after it."""
        if s._push is not None:
            s._push.invalidate()
            s._push = None
        s.submit_pcm16 = s._submit_pcm16_general

    def _input_rate(self, cfg: VADConfig) -> Optional[int]:
        """The rate a session's chunks are resampled from inside the tick, None if it sends the engine's own frames."""
        sr = int(cfg.sample_rate)
        if self.convert_rates and cfg.auto_convert_sample_rate and sr in TICK_RATES:
            return sr
        return None

    def _check_session_rate(self, cfg: VADConfig) -> None:
        sr = self._input_rate(cfg)
        if sr is not None:
            want = FRAME * sr // 16000
            if int(cfg.buffer_size) != want:
                raise AudioProcessingError(f"Failed to resample audio from {sr}Hz to 16000Hz: a chunk must hold {want} samples "
                                           f"(32 ms), the session's buffer_size is {int(cfg.buffer_size)}")
            return
        # V5 + a rate other than 16 kHz: the reference's graph fails on every frame; V4: the pool's engine is
        # one of the graph's two sub-models, a session must ask for the same one
        SileroVADModel._check_rate(cfg.sample_rate, cfg.model_version, self.frame)
        if weights_io.is_8k_variant(4 if cfg.model_version == SileroModelVersion.V4 else 5, cfg.sample_rate) != self._k8:
            raise AudioProcessingError(f"Model prediction failed: this pool runs the {'8' if self._k8 else '16'} kHz "
                                       f"sub-model, the session asks for sample rate {int(cfg.sample_rate)}")

    def close_session(self, s: PooledSession) -> None:
        """Closes the session WHERE IT LIVES: a caller that read ``s.pool`` before a migration finished arrives here with the
        old pool; acting on it would cancel and close whatever stream has taken ``s.slot``'s number in this engine."""
        while True:
            home = s.pool
            if home is not self:
                return home.close_session(s)
            with self._tick_lock:                  # never under a running launch
                with self._lock:
                    if s.closed:
                        return
                    mine = s.pool is self and not s.moving
                    if mine:
                        slot = s.slot
                        self._unbind_push(s)
                        self._flush_inbox()        # then the cancel below takes the session's frames out again
                        s.closed = True
                        s.long_frames.clear()
                        s._held.clear()
                        self._sessions.pop(slot, None)
                        self._by_slot[slot] = None
                        self._cont_cb[slot] = None
                        self._wav_rate[slot] = 0
                if mine:
                    self.engine.tick_cancel(slot)
                    self.engine.close_stream(slot)
                    return
            time.sleep(0.0005)                     # mid-migration: wait for it to land, then close it there

    def reconfigure(self, s: PooledSession, config: VADConfig) -> None:
        """New thresholds / frame length for a live session; like ``ClientState.update_config`` rebuilding its
        wrapper (vad_websocket_server.py:300-318) the stream starts from a clean state."""
        while True:
            home = s.pool
            if home is not self:
                return home.reconfigure(s, config)
            self._check_session_rate(config)
            with self._tick_lock:
                with self._lock:
                    if s.closed:
                        raise AudioProcessingError("session is closed")
                    mine = s.pool is self and not s.moving
                    if mine:
                        self._unbind_push(s)
                        self._flush_inbox()
                        s.long_frames.clear()
                if mine:
                    self.engine.tick_cancel(s.slot)
                    self.engine.reset([s.slot])
                    self.engine.set_thresholds(s.slot, config.vad_start_probability, config.vad_end_probability,
                                               config.voice_start_ratio, config.voice_end_ratio, config.voice_start_frame_count,
                                               config.voice_end_frame_count)
                    s.config = config
                    s.rate = self._input_rate(config)
                    s.gate = bool(config.enable_denoising)
                    contp = bool(self._contp[s.slot])
                    with self._lock:
                        self._init_slot(s.slot, config)
                        # the session keeps its callbacks across a reconfigure (the app binds them once, at open): so does the flag
                        # that makes the tick deliver voice_continue payloads to it
                        self._cont[s.slot], self._contp[s.slot] = s.on_continue is not None, contp
                        self._cont_cb[s.slot] = s.on_continue if not contp else None
                        s.lost = 0
                        self._bind_push(s)
                    s.wav_writer = WAVWriter(sample_rate=config.output_wav_sample_rate, bit_depth=config.output_wav_bit_depth,
                                             channels=1)
                    return
            time.sleep(0.0005)

    @property
    def session_count(self) -> int:
        return len(self._sessions)

    # ------------------------------------------------------------------ ingest
    def submit(self, s: PooledSession, frame) -> None:
        """Queue one frame (float32 in [-1, 1], any length; the model sees it padded / truncated to its frame length)."""
        x = np.asarray(frame)
        if x.dtype != np.float32:
            x = x.astype(np.float32)
        if x.ndim != 1 or x.size == 0:
            raise AudioProcessingError("Audio data cannot be empty" if x.size == 0 else
                                       f"Audio data must be 1D, got {x.ndim}D array")
        if not np.isfinite(x).all():
            raise AudioProcessingError("Audio data contains NaN values" if np.isnan(x).any() else
                                       "Audio data contains infinite values")
        with self._lock:
            if s.closed:
                raise AudioProcessingError("session is closed")
            if s.pool is not self:
                return _RETRY
            if s.moving:
                s._held.append(x)
                return None
            self._ingest_f32(s, x)

    def _ingest_f32(self, s: PooledSession, x: np.ndarray) -> None:
        """``_lock`` held, session in place."""
        self._flush_inbox()                                # frames of one session keep their order across both ingest paths
        self.engine.tick_push(s.slot, x, bool(self._gate[s.slot]), sample_rate=s.rate)
        if x.size > self.frame and s.rate is None:       # only voice_continue payloads need the whole frame here (the engine keeps its own)
            s.long_frames.append(x)

    def submit_pcm16(self, s: PooledSession, data: bytes) -> None:
        """Queue one frame as it arrives on the wire: little-endian int16 PCM.  The bytes go to the GPU as they are
        (half the transfer of float32); the kernel scales by 1/32767 with a true division, which is bit-for-bit what
        the reference server does on the host (vad_websocket_server.py:341).  A frame no longer than the model's only joins
        the pool's inbox here (a list append): the frames that arrived within one tick window reach the engine in one call per
        frame length."""
        if len(data) < 2 or len(data) & 1:
            raise AudioProcessingError("Audio data cannot be empty" if len(data) < 2 else
                                       "PCM16 frame with an odd number of bytes")
        with self._lock:
            if s.closed:
                raise AudioProcessingError("session is closed")
            if s.pool is not self:
                return _RETRY
            if s.moving:
                s._held.append(data)
                return None
            self._ingest_pcm16(s, data)

    def _ingest_pcm16(self, s: PooledSession, data: bytes) -> None:
        """``_lock`` held, session in place."""
        if s.rate is None and len(data) <= 2 * self.frame:
            # the inbox is flushed group by group: a session's frame must not wait there behind a group that holds an
            # EARLIER frame of it (another length: the last chunk of a file), nor behind the C inbox, which is flushed second
            if (self._wire is not None and len(self._wire)) or (s._queued[0] == self._epoch and s._queued[1] != len(data)):
                self._flush_inbox()
            s._queued = (self._epoch, len(data))
            key = (len(data), s.gate)
            box = self._inbox.get(key)
            if box is None:
                box = self._inbox[key] = []
            box.append((s.slot, data))
            return None
        self._flush_inbox()
        self.engine.tick_push(s.slot, data, bool(self._gate[s.slot]), sample_rate=s.rate)
        if len(data) > 2 * self.frame and s.rate is None:
            s.long_frames.append(np.frombuffer(data, dtype="<i2").astype(np.float32) / np.float32(32767.0))

    def _replay_held(self, s: PooledSession) -> None:
        """The frames that arrived while the session moved, in order, onto the pool it landed on (``_lock`` held; called by
        ``ShardedStreamPool.migrate`` right before it clears ``moving``)."""
        while s._held:
            f = s._held.popleft()
            try:
                if isinstance(f, (bytes, bytearray)):
                    self._ingest_pcm16(s, bytes(f))
                else:
                    self._ingest_f32(s, f)
            except Exception as e:
                s.lost += 1
                self._report(s, e if isinstance(e, AudioProcessingError) else AudioProcessingError(f"Model prediction failed: {e}"))

    def _flush_python_inbox(self) -> None:
        self._epoch += 1
        if self._inbox:
            inbox, self._inbox = self._inbox, {}
            for (nbytes, gate), box in inbox.items():
                self._push_joined(nbytes, gate, box)

    def _flush_inbox(self) -> None:
        """The collected wire frames -> the engine's tick staging, one call per gate value (``_lock`` held).  A frame the engine
        refuses (its stream has 256 frames waiting, or was closed meanwhile) is reported to its own session only."""
        self._flush_python_inbox()
        w = self._wire
        if w is not None and len(w):
            if self._wire_entry is not None:
                for slot, status in w.flush(*self._wire_entry()):    # vad_tick_push_gather on the arrays, GIL released
                    self._refused(slot, status)
            else:
                for nbytes, gate, rate, box in w.drain():
                    if rate:
                        for slot, data in box:      # (test doubles: one by one)
                            try:
                                self.engine.tick_push(slot, data, gate, sample_rate=rate)
                            except Exception:
                                self._refused(slot, _ffi.VAD_ERR_BUSY)
                    else:
                        self._push_joined(nbytes, gate, box)

    def _push_joined(self, nbytes: int, gate: bool, box: List) -> None:
        slot_list, datas = zip(*box)
        slots = np.array(slot_list, np.int64)
        # one join + one call (measured at 8 192 sessions: 1.9 ms; an array of 8 192 ctypes pointers for vad_tick_push_gather
        # costs more to build in Python than the join's extra copy - the C inbox builds it in C)
        status = self.engine.tick_push_status(slots, b"".join(datas), nbytes // 2, gate)
        if status.any():
            for i in np.nonzero(status)[0]:
                self._refused(int(slots[i]), int(status[i]))

    def _refused(self, slot: int, status: int) -> None:
        s = self._by_slot[slot] if 0 <= slot < len(self._by_slot) else None
        if s is not None and not s.closed:
            s.lost += 1                        # sent, never to be stepped: the app's back-pressure count must not wait for it
            why = "256 frames are waiting for this stream" if status == _ffi.VAD_ERR_BUSY else f"engine status {status}"
            self._report(s, AudioProcessingError(f"Model prediction failed: frame not queued: {why}"))

    # ------------------------------------------------------------------ the tick
    def tick(self) -> int:
        """Advance every session that has a frame pending by ONE frame — one launch per (wire format, gate) group,
        i.e. one launch when all clients speak the same format (``vad_tick_run``).  Returns the number of frames
        processed.  Callbacks run on the calling thread, in the order the frames were submitted; ticks (and their
        callbacks) never overlap, frames may be submitted while one runs.

        The engine also keeps the segments' audio (``vad_tick_enable_segments``): Python touches a session only on START,
        on END (to wrap the finished segment as WAV) and - if it asked for ``voice_continue`` payloads - while it talks."""
        return conduct_ticks([self])

    def _tick_python(self) -> int:
        """The tick on the calling thread, one engine call after the other (no C inbox / test doubles; ``conduct_ticks`` otherwise)."""
        with self._tick_lock:
            with self._lock:
                self._flush_inbox()
            try:
                # the launches, AND the per-session bookkeeping (last probability, frames done, inside-a-segment) on the pool's
                # arrays, AND the list of entries somebody has to hear about - in one C call with the GIL released
                res = self.engine.tick_run_work(0.01, self._lastp, self._done, self._active, self._cont, self._contp)
            except Exception as e:
                return self._tick_failed(e)
            return self._fan_out(res, None)

    def _tick_failed(self, e: Exception) -> int:
        # engine failure: the tick's frames are gone (the engine has dropped what belonged to them, so every stream's
        # queue is still aligned); the sessions that lost a frame hear about it, and so does everyone if the engine
        # cannot say who it was
        lost = getattr(self.engine, "last_tick_lost", None)
        err = AudioProcessingError(f"Model prediction failed: {e}")
        if lost is None:
            victims = list(self._sessions.values())
        else:
            victims = []
            for slot, L in zip(lost[0].tolist(), lost[1].tolist()):
                s = self._by_slot[slot] if slot < len(self._by_slot) else None
                if s is None or s.closed:
                    continue
                if L > self.frame and s.rate is None and s.long_frames:
                    s.long_frames.popleft()          # the whole over-long frame kept for voice_continue went with it
                victims.append(s)
        for s in victims:
            s.lost += 1
            self._report(s, err)
        self.backlog = 0                   # a failing engine is not hammered back to back: the next tick waits its interval
        return 0

    def _fan_out(self, res, wavs) -> int:
        """What a tick produced -> the sessions' callbacks (``_tick_lock`` held).  ``wavs``: {work entry: voice_end payload} for the
        segments whose payload the conductor has built already, or None."""
        slots, gs, frames, nsamp, widx, wkind, wsamp = res
        n = int(slots.size)
        self.backlog = int(getattr(self.engine, "last_tick_staged_next", 0))    # > 0: someone is ahead of the ticker
        if n == 0:
            return 0
        # launches: one per (format, gate) group; the resampled groups of a gate value share one
        self.launches += sum(1 for g in range(6) if gs[g + 1] > gs[g]) + int(gs[9] > gs[6]) + int(gs[12] > gs[9])
        self.ticks += 1
        self.frames += n
        if widx.size == 0:                      # idle and silently talking sessions cost no Python at all
            return n
        # the sessions with something to hear, in the order their frames were stepped.  kind: VAD_WORK_* bits
        START, END, CONT, PAYLOAD, LONG = (_ffi.VAD_WORK_START, _ffi.VAD_WORK_END, _ffi.VAD_WORK_CONTINUE, _ffi.VAD_WORK_PAYLOAD,
                                           _ffi.VAD_WORK_LONG)
        by_slot = self._by_slot
        gs_l = None
        wslots = slots[widx]
        entry = np.arange(widx.size)            # position in the work list (the conductor's payloads are keyed by it)
        if _wirebox is not None:
            # the voice_continue NOTIFICATIONS that come alone (no START / END / payload on the same frame) - the reference
            # protocol sends one per frame and talking client - are delivered by one C loop over the callbacks
            plain = wkind == CONT
            if plain.any():
                for slot, exc in _wirebox.call_each(self._cont_cb, np.ascontiguousarray(wslots[plain]), b""):
                    s = by_slot[slot]
                    if s is not None:
                        self._report(s, CallbackError("voice_continue", exc))
                rest = ~plain
                wslots, wkind, widx, wsamp, entry = wslots[rest], wkind[rest], widx[rest], wsamp[rest], entry[rest]
        for slot, kind, i, seg_n, j in zip(wslots.tolist(), wkind.tolist(), widx.tolist(), wsamp.tolist(), entry.tolist()):
            s = by_slot[slot]
            if s is None or s.closed:
                continue
            if kind == CONT:                    # (a lone notification: only without the C inbox, see above)
                cb = s.on_continue
                if cb is not None:
                    try:
                        cb(b"")
                    except Exception as e:
                        self._report(s, CallbackError("voice_continue", e))
                continue
            try:
                whole = s.long_frames.popleft() if (kind & LONG and s.long_frames) else None
                if kind & START:
                    self._call(s.on_start, "voice_start")
                # order on the END frame as the reference's wrapper delivers it: voice_end, then voice_continue
                # (core/vad_wrapper.py:505-519)
                if kind & END:
                    wav = wavs.get(j) if wavs is not None else None
                    self._call(s.on_end, "voice_end", wav if wav is not None else self._segment_wav(s, slot, seg_n))
                if kind & CONT and s.on_continue is not None:
                    if not kind & PAYLOAD:
                        self._call(s.on_continue, "voice_continue", b"")          # a notification: no payload is built
                    else:
                        if gs_l is None:
                            gs_l = gs.tolist()
                        g = bisect.bisect_right(gs_l, i, 1) - 1                       # the entry's (format, gate) group
                        if whole is None:
                            x = frames[g][i - gs_l[g]][:int(nsamp[i])]
                            whole = x.astype(np.float32) / np.float32(32767.0 if g < 4 else 32768.0) if 2 <= g < 6 else x.copy()
                        if (g & 1) if g < 6 else g >= 9:
                            whole = AudioUtils.denoise_audio(whole)
                        self._call(s.on_continue, "voice_continue", whole.tobytes())
            except Exception as e:
                self._report(s, e)
        return n

    def _segment_wav(self, s: PooledSession, slot: int, nsamples: int) -> bytes:
        """The finished segment as the voice_end payload.  16-bit mono - every session unless configured otherwise - is written
        by the engine (vad_tick_take_segment_wav16: the same bytes as WAVWriter's, without a trip through numpy), with the C
        inbox built straight into the bytes object."""
        ww = s.wav_writer
        if ww.bit_depth == 16 and ww.channels == 1:
            if self._wav_entry is not None:
                return _wirebox.take_wav16(*self._wav_entry(), slot, ww.sample_rate, nsamples)
            return self.engine.tick_take_segment_wav16(slot, ww.sample_rate)
        return ww.write_wav_data(self.engine.tick_take_segment(slot))

    @staticmethod
    def _report(s: PooledSession, e: Exception) -> None:
        """A session's error hook must not take the ticker down with it (a closing event loop raises in call_soon_threadsafe)."""
        if s.on_error is None:
            return
        try:
            s.on_error(e)
        except Exception:
            pass

    @staticmethod
    def _call(cb, name: str, *args) -> None:
        if cb is None:
            return
        try:
            cb(*args)
        except Exception as e:
            raise CallbackError(name, e)

    # ------------------------------------------------------------------ tickers
    def drain(self, max_ticks: int = 1 << 30) -> int:
        """Tick until no session has a pending frame (tests, offline use)."""
        total = 0
        for _ in range(max_ticks):
            n = self.tick()
            if n == 0:
                break
            total += n
        return total

    def start(self) -> None:
        """Background ticker thread: one tick every ``tick_interval`` seconds (sooner never helps: a client sends
        one frame per 30 ms)."""
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
                if self.backlog:                    # a client is ahead of real time (a file, a stall that ended): catch up, do not sleep
                    continue
                self._stop.wait(max(0.0, self.tick_interval - (time.perf_counter() - t0)))

        self._thread = threading.Thread(target=loop, name="vad-pool-ticker", daemon=True)
        self._thread.start()

    def stop(self) -> None:
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def close(self) -> None:
        self.stop()
        for s in list(self._sessions.values()):
            self.close_session(s)

    def stats(self) -> dict:
        return {"sessions": self.session_count, "ticks": self.ticks, "frames": self.frames, "launches": self.launches,
                "frames_per_launch": self.frames / self.launches if self.launches else 0.0}


def conduct_ticks(pools) -> int:
    """ONE tick of every pool in ``pools`` (one pool per GPU), conducted by the calling thread: the engines' share - inbox hand-over,
    launches, bookkeeping, the finished segments' WAV payloads - runs side by side on C threads behind a single release of the
    interpreter lock (``_wirebox.tick_shards``), then this thread fans the events out pool by pool.  With a Python ticker thread
    per pool the threads spent the round queueing for that lock (65 536 sessions on 8 pools: 0.4 x real time).
    Pools without the C inbox (or on test doubles) tick one after the other through ``SharedStreamPool._tick_python``.
    -> frames processed."""
    pools = list(pools)
    fast = _wirebox is not None and all(p._wire is not None and p._wire_entry is not None
                                        and hasattr(p.engine, "tick_work_begin") for p in pools)
    if not fast:
        return sum(p._tick_python() for p in pools)
    order = sorted(pools, key=id)
    for p in order:
        p._tick_lock.acquire()
    try:
        jobs, structs = [], []
        for p in order:
            p._lock.acquire()                   # frames on the general path keep their place relative to the inbox's (see submit*)
        try:
            for p in pools:
                p._flush_python_inbox()
                r, w = p.engine.tick_work_begin(p._lastp, p._done, p._active, p._cont, p._contp)
                structs.append((r, w))
                fn, eng, rate_fn = p._wire_entry()
                wav = p._wav_entry() if p._wav_entry is not None else (0, eng)
                jobs.append((p._wire, fn, eng, rate_fn or 0, p.engine.tick_work_entry(), 0.01, C.addressof(r), C.addressof(w),
                             wav[0], p._wav_rate if wav[0] else None))
            out = _wirebox.tick_shards(jobs)
        finally:
            for p in order:
                p._lock.release()
        total = 0
        for p, (r, w), (rc, fails, wavs) in zip(pools, structs, out):
            for slot, status in fails:
                p._refused(slot, status)
            try:
                res = p.engine.tick_work_end(r, w, rc)
            except Exception as e:
                total += p._tick_failed(e)
                continue
            total += p._fan_out(res, dict(wavs) if wavs else None)
        return total
    finally:
        for p in order:
            p._tick_lock.release()

"""Python face of the C ABI (``include/vad_engine.h``): one :class:`Engine` = one GPU's stream pool.

This is the multi-stream operator the reference lacks.  The reference owns one
``SileroVADModel`` (one ORT session + one ``(h, c)``) per client
(/root/reference/websocket_service/server/vad_websocket_server.py:277,
/root/reference/src/real_time_vad/core/silero_model.py:238-566); here many streams share one
engine and advance together in one kernel launch.  Errors are mapped onto the reference's
exception classes with the same message prefixes (SURVEY §8 b).
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .core.exceptions import AudioProcessingError, ModelInitializationError, VADError

_FMT = {np.dtype(np.float32): _ffi.VAD_FMT_F32, np.dtype(np.int16): _ffi.VAD_FMT_I16_32767}
TICK_GROUPS = 12                      # include/vad_engine.h VAD_TICK_GROUPS
TICK_RATES = (8000, 24000, 48000)     # tick groups 6.. : 6 + 3 * gate_on + index into this
TICK_RATE_CHUNK = (256, 768, 1536)    # samples of such a chunk = one 16 kHz frame's worth


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Engine:
    """Stream pool + fused Silero kernels on one MI355X.

    ``weights`` is an SVW blob (``weights_io.load_weight_blob``).  ``denoise`` is the gate
    threshold of ``AudioUtils.denoise_audio`` (0.01) or ``None`` to disable it.
    """

    def __init__(self, weights: bytes, model_version: int = 5, device_id: int = 0, max_streams: int = 8192,
                 sample_rate: int = 16000, shared_gpu: bool = False):
        self._lib = _ffi.lib()
        self._gather_fn = int(C.cast(self._lib.vad_tick_push_gather, C.c_void_p).value)
        self._rate_gather_fn = int(C.cast(self._lib.vad_tick_push_rate_gather, C.c_void_p).value)
        self._h = C.c_void_p()
        self._tickets = {}                  # ticket -> the buffers a pipelined call still reads (submit / collect)
        self.last_tick_us = (0.0, 0.0, 0.0)
        self.last_tick_dropped = 0
        self.last_tick_staged_next = 0
        self.last_tick_lost = None
        self._weights = weights  # keep alive during create
        desc = _ffi.EngineDesc(C.sizeof(_ffi.EngineDesc), model_version, C.cast(C.c_char_p(weights), C.c_void_p),
                               len(weights), device_id, max_streams, sample_rate, 1 if shared_gpu else 0)  # VAD_ENGINE_SHARED_GPU
        rc = self._lib.vad_engine_create(C.byref(desc), C.byref(self._h))
        if rc != _ffi.VAD_OK:
            self._h = C.c_void_p()
            msg = self._lib.vad_last_create_error().decode() or f"vad_engine_create failed ({rc})"
            raise ModelInitializationError(f"v{model_version}", msg)
        self.model_version = model_version
        self.max_streams = max_streams
        self.device_id = device_id
        self.sample_rate = sample_rate
        self.frame_samples = 256 if (model_version == 5 and sample_rate != 16000) else 512   # vad_info.frame_samples

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        h, self._h = self._h, C.c_void_p()
        if h:
            self._lib.vad_engine_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, exc=AudioProcessingError) -> None:
        if rc != _ffi.VAD_OK:
            msg = self._lib.vad_last_error(self._h).decode() or f"engine call failed ({rc})"
            raise exc(msg)

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def info(self) -> dict:
        inf = _ffi.EngineInfo()
        inf.struct_size = C.sizeof(_ffi.EngineInfo)
        self._check(self._lib.vad_engine_info(self._h, C.byref(inf)), VADError)
        out = {k: getattr(inf, k) for k, _ in _ffi.EngineInfo._fields_ if k != "struct_size"}
        out["device_name"] = inf.device_name.decode()
        out["arch"] = inf.arch.decode()
        return out

    def set_tile(self, streams_per_tile: int = 0) -> None:
        """Diagnostic (``vad_debug_set_tile``): 0 = pick the kernel shape by batch size, 16 / 32 = force it."""
        self._check(self._lib.vad_debug_set_tile(self._h, int(streams_per_tile)), VADError)

    def synchronize(self) -> None:
        self._check(self._lib.vad_engine_synchronize(self._h))

    def pinned_array(self, shape, dtype=np.float32) -> np.ndarray:
        """A numpy array over page-locked host memory (``vad_host_alloc``): frame and result buffers handed to the
        host-pointer entry points from it are DMA'd without a staging copy.  The memory belongs to the engine and is
        released by ``close()``: do not use the array after that."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = C.c_void_p()
        self._check(self._lib.vad_host_alloc(self._h, max(1, n * dt.itemsize), C.byref(p)))
        buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    # ------------------------------------------------------------------ streams
    def open_stream(self) -> int:
        s = C.c_int64()
        self._check(self._lib.vad_stream_open(self._h, C.byref(s)), VADError)
        return int(s.value)

    def open_streams(self, n: int) -> np.ndarray:
        out = np.empty(int(n), np.int64)
        self._check(self._lib.vad_stream_open_many(self._h, int(n), _ptr(out, C.c_int64)), VADError)
        return out

    def close_stream(self, slot: int) -> None:
        self._check(self._lib.vad_stream_close(self._h, int(slot)), VADError)

    def reset(self, slots: Sequence[int]) -> None:
        s = np.ascontiguousarray(slots, dtype=np.int64)
        self._check(self._lib.vad_stream_reset(self._h, _ptr(s, C.c_int64), s.size), VADError)

    def get_state(self, slot: int) -> np.ndarray:
        out = np.empty(_ffi.VAD_STATE_FLOATS, np.float32)
        self._check(self._lib.vad_stream_get_state(self._h, int(slot), _ptr(out, C.c_float)), VADError)
        return out

    def set_state(self, slot: int, hc: np.ndarray) -> None:
        hc = np.ascontiguousarray(hc, np.float32).reshape(_ffi.VAD_STATE_FLOATS)
        self._check(self._lib.vad_stream_set_state(self._h, int(slot), _ptr(hc, C.c_float)), VADError)

    def save_stream(self, slot: int) -> bytes:
        """(h, c) + state machine of one stream as an opaque blob (``vad_stream_save``)."""
        buf = C.create_string_buffer(_ffi.VAD_STREAM_SAVE_BYTES)
        self._check(self._lib.vad_stream_save(self._h, int(slot), buf, _ffi.VAD_STREAM_SAVE_BYTES), VADError)
        return buf.raw

    def restore_stream(self, slot: int, blob: bytes) -> None:
        self._check(self._lib.vad_stream_restore(self._h, int(slot), blob, len(blob)), VADError)

    def set_thresholds(self, slot: int, start_probability=0.7, end_probability=0.7, start_ratio=0.8, end_ratio=0.95,
                       start_frame_count=10, end_frame_count=50) -> None:
        t = _ffi.Thresholds(start_probability, end_probability, start_ratio, end_ratio, start_frame_count,
                            end_frame_count)
        self._check(self._lib.vad_stream_set_thresholds(self._h, int(slot), C.byref(t)), VADError)

    def set_thresholds_many(self, slots, thresholds) -> None:
        """``thresholds``: one 6-tuple (shared by all slots) or one per slot, in the order of ``set_thresholds``' arguments
        (``vad_stream_set_thresholds_many``: one launch for the lot)."""
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        rows = [thresholds] if np.isscalar(thresholds[0]) else list(thresholds)
        arr = (_ffi.Thresholds * len(rows))(*[_ffi.Thresholds(*r) for r in rows])
        self._check(self._lib.vad_stream_set_thresholds_many(self._h, _ptr(s, C.c_int64), s.size, arr, len(rows)), VADError)

    def debug_sm_replay(self, slot: int, probs) -> Tuple[np.ndarray, np.ndarray]:
        """Diagnostic: run scripted probabilities through one slot's device state machine."""
        p = np.ascontiguousarray(probs, np.float32)
        ev = np.zeros(p.size, np.uint8)
        seg = np.zeros(p.size, np.int32)
        self._check(self._lib.vad_debug_sm_replay(self._h, int(slot), _ptr(p, C.c_float), p.size, _ptr(ev, C.c_uint8),
                                                  _ptr(seg, C.c_int32)), VADError)
        return ev, seg

    # ------------------------------------------------------------------ hot path
    def _prep(self, slots, frames, T: Optional[int]) -> Tuple[np.ndarray, np.ndarray, int]:
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        f = np.asarray(frames)
        if f.dtype not in _FMT:
            f = f.astype(np.float32)
        f = np.ascontiguousarray(f)
        want = (s.size, self.frame_samples) if T is None else (s.size, T, self.frame_samples)
        if f.shape != want:
            raise AudioProcessingError(f"Model prediction failed: frames have shape {f.shape}, expected {want}")
        return s, f, _FMT[f.dtype]

    def step(self, slots, frames, denoise: Optional[float] = 0.01, i16_scale: int = 32767) -> np.ndarray:
        """One 512-sample frame per listed stream -> probabilities [n]."""
        s, f, fmt = self._prep(slots, frames, None)
        if fmt != _ffi.VAD_FMT_F32 and i16_scale == 32768:
            fmt = _ffi.VAD_FMT_I16_32768
        probs = np.empty(s.size, np.float32)
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step(self._h, _ptr(s, C.c_int64), s.size, f.ctypes.data_as(C.c_void_p), fmt, thr,
                                       _ptr(probs, C.c_float)))
        return probs

    def step_events(self, slots, frames, denoise: Optional[float] = 0.01, i16_scale: int = 32767):
        """-> (probs [n], event bits [n] uint8, finished-segment frames [n] int32)."""
        s, f, fmt = self._prep(slots, frames, None)
        if fmt != _ffi.VAD_FMT_F32 and i16_scale == 32768:
            fmt = _ffi.VAD_FMT_I16_32768
        probs = np.empty(s.size, np.float32)
        ev = np.zeros(s.size, np.uint8)
        seg = np.zeros(s.size, np.int32)
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step_events(self._h, _ptr(s, C.c_int64), s.size, f.ctypes.data_as(C.c_void_p), fmt,
                                              thr, _ptr(probs, C.c_float), _ptr(ev, C.c_uint8), _ptr(seg, C.c_int32)))
        return probs, ev, seg

    def step_multi(self, slots, frames, denoise: Optional[float] = 0.01, i16_scale: int = 32767):
        """frames [n, T, 512]: T consecutive frames per stream -> (probs [n,T], events [n,T])."""
        f0 = np.asarray(frames)
        if f0.ndim != 3:
            raise AudioProcessingError(f"Model prediction failed: frames must be [n, T, 512], got {f0.shape}")
        T = int(f0.shape[1])
        s, f, fmt = self._prep(slots, f0, T)
        if fmt != _ffi.VAD_FMT_F32 and i16_scale == 32768:
            fmt = _ffi.VAD_FMT_I16_32768
        probs = np.empty((s.size, T), np.float32)
        ev = np.zeros((s.size, T), np.uint8)
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step_multi(self._h, _ptr(s, C.c_int64), s.size, T, f.ctypes.data_as(C.c_void_p), fmt,
                                             thr, _ptr(probs, C.c_float), _ptr(ev, C.c_uint8)))
        return probs, ev

    def step_device(self, n: int, d_frames: int, d_probs: int, d_slots: int = 0, d_events: int = 0, d_seg: int = 0,
                    fmt: int = _ffi.VAD_FMT_F32, denoise: Optional[float] = 0.01, stream: int = 0) -> None:
        """Asynchronous launch on device pointers (integers, e.g. ``tensor.data_ptr()``)."""
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step_device(self._h, d_slots or None, n, d_frames, fmt, thr, d_probs,
                                              d_events or None, d_seg or None, stream or None))

    def step_multi_device(self, n: int, T: int, d_frames: int, d_probs: int, d_slots: int = 0, d_events: int = 0,
                          d_seg: int = 0, fmt: int = _ffi.VAD_FMT_F32, denoise: Optional[float] = 0.01, stream: int = 0) -> None:
        """``step_device`` with T frames per stream: d_frames [n, T, frame], d_probs [n, T] (``vad_step_multi_device``)."""
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step_multi_device(self._h, d_slots or None, n, T, d_frames, fmt, thr, d_probs,
                                                    d_events or None, d_seg or None, stream or None))

    # ------------------------------------------------------------------ tick assembler (shared-pool serving)
    def tick_push(self, slot: int, frame, gate_on: bool = True, i16_scale: int = 32767, sample_rate: Optional[int] = None) -> None:
        """Queue one frame for ``slot`` (``vad_tick_push``): ``bytes`` = little-endian int16 PCM as it came off the wire,
        or a float32 array.  Written straight into the coming tick's page-locked staging; padded / truncated to the
        model's frame length.  ``sample_rate`` other than the engine's (8000 / 24000 / 48000 on a 16 kHz engine): the chunk
        that yields one frame, resampled on the GPU inside the tick (``vad_tick_push_rate``)."""
        i16_fmt = _ffi.VAD_FMT_I16_32768 if i16_scale == 32768 else _ffi.VAD_FMT_I16_32767
        if isinstance(frame, (bytes, bytearray, memoryview)):
            buf, count, fmt = bytes(frame), len(frame) // 2, i16_fmt
        else:
            f = np.ascontiguousarray(frame)
            fmt = i16_fmt
            if f.dtype != np.int16:
                f = np.ascontiguousarray(f, np.float32)
                fmt = _ffi.VAD_FMT_F32
            buf, count = f.ctypes.data_as(C.c_void_p), f.size
        if sample_rate is None or int(sample_rate) == self.sample_rate:
            self._check(self._lib.vad_tick_push(self._h, int(slot), buf, count, fmt, int(gate_on)))
        else:
            self._check(self._lib.vad_tick_push_rate(self._h, int(slot), buf, count, fmt, int(gate_on), int(sample_rate)))

    def tick_push_many(self, slots, frames, gate_on: bool = True, i16_scale: int = 32767) -> None:
        """frames [n, L] (float32 or int16), one per listed slot (``vad_tick_push_many``)."""
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        f = np.ascontiguousarray(frames)
        if f.dtype == np.int16:
            fmt = _ffi.VAD_FMT_I16_32768 if i16_scale == 32768 else _ffi.VAD_FMT_I16_32767
        else:
            f = np.ascontiguousarray(f, np.float32)
            fmt = _ffi.VAD_FMT_F32
        if f.ndim != 2 or f.shape[0] != s.size:
            raise AudioProcessingError(f"Model prediction failed: frames have shape {f.shape}, expected ({s.size}, L)")
        self._check(self._lib.vad_tick_push_many(self._h, _ptr(s, C.c_int64), s.size, f.ctypes.data_as(C.c_void_p), f.shape[1],
                                                 fmt, int(gate_on)))

    def tick_enable_segments(self, on: bool = True) -> None:
        self._check(self._lib.vad_tick_enable_segments(self._h, int(on)), VADError)

    def tick_take_segment(self, slot: int) -> np.ndarray:
        """The finished segment of ``slot`` as float32 samples (``vad_tick_take_segment``); empty if there is none."""
        n = C.c_int64()
        self._check(self._lib.vad_tick_take_segment(self._h, int(slot), None, 0, C.byref(n)), VADError)
        out = np.empty(int(n.value), np.float32)
        self._check(self._lib.vad_tick_take_segment(self._h, int(slot), _ptr(out, C.c_float), out.size, C.byref(n)), VADError)
        return out

    def tick_push_status(self, slots, frames, nsamples: int, gate_on: bool = True, i16_scale: int = 32767) -> np.ndarray:
        """``frames``: int16 / float32 array [n, nsamples] or the same as one bytes object of int16 PCM; every frame is tried,
        -> int32 status per frame (``vad_tick_push_status``; 0 = queued)."""
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        if isinstance(frames, (bytes, bytearray, memoryview)):
            buf, fmt = frames, (_ffi.VAD_FMT_I16_32768 if i16_scale == 32768 else _ffi.VAD_FMT_I16_32767)
            if len(frames) != 2 * int(nsamples) * s.size:
                raise AudioProcessingError(f"Model prediction failed: {len(frames)} bytes for {s.size} int16 frames of {nsamples} samples")
            ptr = C.cast(C.c_char_p(bytes(frames) if not isinstance(frames, bytes) else frames), C.c_void_p)
        else:
            f = np.ascontiguousarray(frames)
            if f.dtype == np.int16:
                fmt = _ffi.VAD_FMT_I16_32768 if i16_scale == 32768 else _ffi.VAD_FMT_I16_32767
            else:
                f, fmt = np.ascontiguousarray(f, np.float32), _ffi.VAD_FMT_F32
            if f.size != int(nsamples) * s.size:
                raise AudioProcessingError(f"Model prediction failed: frames {f.shape} for {s.size} slots of {nsamples} samples")
            buf, ptr = f, f.ctypes.data_as(C.c_void_p)
        status = np.zeros(s.size, np.int32)
        self._lib.vad_tick_push_status(self._h, _ptr(s, C.c_int64), s.size, ptr, int(nsamples), fmt, int(gate_on), _ptr(status, C.c_int32))
        del buf
        return status

    def tick_push_gather(self, slots, frames, nsamples: int, gate_on: bool = True, i16_scale: int = 32767) -> np.ndarray:
        """``frames``: a sequence of ``bytes`` objects (int16 PCM, ``nsamples`` samples each), one per listed slot; they are
        copied from where they are into the tick's staging (``vad_tick_push_gather``) -> int32 status per frame."""
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        n = s.size
        if len(frames) != n:
            raise AudioProcessingError(f"Model prediction failed: {len(frames)} frames for {n} slots")
        ptrs = (C.c_char_p * n)(*frames)               # borrows the bytes objects' buffers: `frames` stays alive through the call
        status = np.zeros(n, np.int32)
        fmt = _ffi.VAD_FMT_I16_32768 if i16_scale == 32768 else _ffi.VAD_FMT_I16_32767
        self._lib.vad_tick_push_gather(self._h, _ptr(s, C.c_int64), n, ptrs, int(nsamples), fmt, int(gate_on), _ptr(status, C.c_int32))
        return status

    def tick_push_rate_gather(self, slots, frames, sample_rate: int, gate_on: bool = True, i16_scale: int = 32767) -> np.ndarray:
        """``frames``: a sequence of ``bytes`` objects, one int16 chunk of ``512 * sample_rate / 16000`` samples per listed slot, all
        at ONE input rate (``vad_tick_push_rate_gather``) -> int32 status per chunk."""
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        n = s.size
        if len(frames) != n:
            raise AudioProcessingError(f"Model prediction failed: {len(frames)} frames for {n} slots")
        nsamples = len(frames[0]) // 2 if n else 0
        if any(len(f) != 2 * nsamples for f in frames):
            raise AudioProcessingError("Model prediction failed: chunks of one call must have one length")
        ptrs = (C.c_char_p * n)(*frames)
        status = np.zeros(n, np.int32)
        fmt = _ffi.VAD_FMT_I16_32768 if i16_scale == 32768 else _ffi.VAD_FMT_I16_32767
        self._lib.vad_tick_push_rate_gather(self._h, _ptr(s, C.c_int64), n, ptrs, int(nsamples), fmt, int(gate_on), int(sample_rate),
                                            _ptr(status, C.c_int32))
        return status

    def tick_gather_entry(self):
        """(address of ``vad_tick_push_gather``, address of this engine, address of ``vad_tick_push_rate_gather``) for callers
        that push from C (server/_wirebox)."""
        if not self._h:
            raise VADError("engine is closed")
        return self._gather_fn, int(self._h.value), self._rate_gather_fn

    def tick_cancel(self, slot: int) -> None:
        self._check(self._lib.vad_tick_cancel(self._h, int(slot)), VADError)

    def tick_pending(self, slot: int) -> int:
        """frames of ``slot`` that have been pushed and not stepped yet (``vad_tick_pending``)"""
        k = C.c_int64()
        self._check(self._lib.vad_tick_pending(self._h, int(slot), C.byref(k)), VADError)
        return int(k.value)

    def save_segment(self, slot: int) -> bytes:
        """the slot's segment audio (pre-roll, open segment, finished one not yet taken) as an opaque blob
        (``vad_tick_segment_save``): with ``save_stream`` everything a session needs to continue on another engine"""
        k = C.c_int64()
        self._check(self._lib.vad_tick_segment_save(self._h, int(slot), None, 0, C.byref(k)), VADError)
        buf = C.create_string_buffer(int(k.value))
        self._check(self._lib.vad_tick_segment_save(self._h, int(slot), buf, int(k.value), C.byref(k)), VADError)
        return buf.raw[:int(k.value)]

    def restore_segment(self, slot: int, blob: bytes) -> None:
        self._check(self._lib.vad_tick_segment_restore(self._h, int(slot), blob, len(blob)), VADError)

    def tick_run(self, denoise: float = 0.01):
        """Advance every slot with a pending frame by one frame (``vad_tick_run``) ->
        ``(slots, probs, events, seg_frames, group_start, frames, nsamples)``: arrays over the stepped streams (views of
        engine-owned page-locked memory, valid until the next ``tick_run``), ``group_start`` [13], ``frames[g]`` = the staged
        audio of group g as an array [count, frame] or None - g = fmt * 2 + gate_on for g < 6 (float32 for g < 2, int16
        above), g = 6 + 3 * gate_on + {0: 8 kHz, 1: 24 kHz, 2: 48 kHz} for chunks at another rate (float32 [count, 256 / 768 /
        1536]) - and ``nsamples`` = the length each frame had when it was pushed."""
        r = _ffi.TickResult()
        r.struct_size = C.sizeof(_ffi.TickResult)
        rc = self._lib.vad_tick_run(self._h, float(denoise), C.byref(r))
        return self._tick_arrays(r, rc)

    def _tick_arrays(self, r, rc):
        self.last_tick_dropped = int(r.dropped)   # frames left out because their stream was closed after the push
        self.last_tick_staged_next = int(r.staged_next)   # frames that were waiting and are already staged for the next tick
        # a failed tick has consumed its frames: which streams lost one, and how long those frames were (TickFailure below)
        self.last_tick_lost = None
        if rc != _ffi.VAD_OK:
            if int(r.n) and r.slots and r.nsamples:
                self.last_tick_lost = (np.ctypeslib.as_array(r.slots, (int(r.n),)).copy(), np.ctypeslib.as_array(r.nsamples, (int(r.n),)).copy())
            self._check(rc)
        n = int(r.n)
        self.last_tick_us = tuple(r.host_us)       # (swap, copies + launches + wait, segment assembly) of this tick
        gs = np.array(list(r.group_start), np.int64)
        if n == 0:
            e = np.empty(0)
            return (e.astype(np.int64), e.astype(np.float32), e.astype(np.uint8), e.astype(np.int32), gs, [None] * TICK_GROUPS,
                    e.astype(np.int32))
        as_arr = np.ctypeslib.as_array
        slots, probs = as_arr(r.slots, (n,)), as_arr(r.probs, (n,))
        events, seg = as_arr(r.events, (n,)), as_arr(r.seg_frames, (n,))
        frames = []
        for g in range(TICK_GROUPS):
            cnt = int(gs[g + 1] - gs[g])
            if cnt == 0 or not r.group_frames[g]:
                frames.append(None)
                continue
            ct = C.c_int16 if 2 <= g < 6 else C.c_float
            flen = self.frame_samples if g < 6 else TICK_RATE_CHUNK[(g - 6) % 3]
            frames.append(as_arr(C.cast(r.group_frames[g], C.POINTER(ct)), (cnt, flen)))
        return slots, probs, events, seg, gs, frames, as_arr(r.nsamples, (n,))

    def tick_run_work(self, denoise: float, last_prob: np.ndarray, frames_done: np.ndarray, active: np.ndarray,
                      continue_cb: np.ndarray, continue_payload: np.ndarray):
        """``tick_run`` + the per-session bookkeeping of a serving front end in the same C call (``vad_tick_run_work``): the five
        per-slot arrays (float32, int64, bool, bool, bool; the caller's, updated in place) and ->
        ``(slots, group_start, frames, nsamples, work_index, work_kind, work_samples)``: ``work_index[j]`` = the entry of the tick's
        arrays the caller has something to do for, ``work_kind[j]`` = VAD_WORK_* bits (START, END, CONTINUE, PAYLOAD, LONG),
        ``work_samples[j]`` = the finished segment's length for END entries."""
        r, w = self.tick_work_begin(last_prob, frames_done, active, continue_cb, continue_payload)
        return self.tick_work_end(r, w, self._lib.vad_tick_run_work(self._h, float(denoise), C.byref(r), C.byref(w)))

    def tick_work_begin(self, last_prob: np.ndarray, frames_done: np.ndarray, active: np.ndarray, continue_cb: np.ndarray,
                        continue_payload: np.ndarray):
        """The two structs of a ``vad_tick_run_work`` call, inputs filled in - for callers that make the call themselves
        (``_wirebox.tick_shards``: several engines' ticks side by side); ``tick_work_end`` turns them into arrays afterwards."""
        r = _ffi.TickResult()
        r.struct_size = C.sizeof(_ffi.TickResult)
        w = _ffi.TickWork()
        w.struct_size = C.sizeof(_ffi.TickWork)
        w.n_slots = int(last_prob.size)
        assert frames_done.size == w.n_slots and active.size == w.n_slots and continue_cb.size == w.n_slots and continue_payload.size == w.n_slots
        w.last_prob = _ptr(last_prob, C.c_float)
        w.frames_done = _ptr(frames_done, C.c_int64)
        w.active = active.ctypes.data_as(C.POINTER(C.c_uint8))
        w.continue_cb = continue_cb.ctypes.data_as(C.POINTER(C.c_uint8))
        w.continue_payload = continue_payload.ctypes.data_as(C.POINTER(C.c_uint8))
        return r, w

    def tick_work_end(self, r, w, rc: int):
        slots, _p, _ev, _seg, gs, frames, nsamp = self._tick_arrays(r, rc)
        nw = int(w.n_work)
        if nw == 0:
            return slots, gs, frames, nsamp, np.empty(0, np.int32), np.empty(0, np.uint8), np.empty(0, np.int64)
        as_arr = np.ctypeslib.as_array
        return slots, gs, frames, nsamp, as_arr(w.work_index, (nw,)), as_arr(w.work_kind, (nw,)), as_arr(w.work_samples, (nw,))

    def tick_work_entry(self) -> int:
        """address of ``vad_tick_run_work``"""
        return C.cast(self._lib.vad_tick_run_work, C.c_void_p).value

    def tick_wav_entry(self):
        """(address of ``vad_tick_take_segment_wav16``, address of this engine) for callers that take segments from C
        (``_wirebox.take_wav16``: the payload is written straight into the bytes object)."""
        if not self._h:
            raise VADError("engine is closed")
        return C.cast(self._lib.vad_tick_take_segment_wav16, C.c_void_p).value, int(self._h.value)

    def tick_take_segment_wav16(self, slot: int, sample_rate: int) -> bytes:
        """The finished segment of ``slot`` as the ``voice_end_callback`` payload: RIFF/WAVE header + int16 PCM, built in the
        engine (``vad_tick_take_segment_wav16``), byte for byte ``WAVWriter(sample_rate, 16, 1).write_wav_data`` of it."""
        n = C.c_int64()
        self._check(self._lib.vad_tick_take_segment_wav16(self._h, int(slot), int(sample_rate), None, 0, C.byref(n)), VADError)
        buf = bytearray(int(n.value))
        self._check(self._lib.vad_tick_take_segment_wav16(self._h, int(slot), int(sample_rate),
                                                          (C.c_char * len(buf)).from_buffer(buf), len(buf), C.byref(n)), VADError)
        return bytes(buf)

    # ------------------------------------------------------------------ pipelined host ingest
    def submit(self, slots, frames, denoise: Optional[float] = 0.01, i16_scale: int = 32767) -> int:
        """Enqueue copy-in -> step -> copy-out for ``frames`` [n, frame] or [n, T, frame] and return a ticket
        (``vad_step_submit``).  ``slots`` / ``frames`` must stay alive and unchanged until ``collect(ticket)``; frames in a
        ``pinned_array`` are DMA'd asynchronously, so the copy of this ticket overlaps the kernel of the previous one."""
        f0 = np.asarray(frames)
        T = int(f0.shape[1]) if f0.ndim == 3 else 1
        s, f, fmt = self._prep(slots, f0, T if f0.ndim == 3 else None)
        if fmt != _ffi.VAD_FMT_F32 and i16_scale == 32768:
            fmt = _ffi.VAD_FMT_I16_32768
        thr = -1.0 if denoise is None else float(denoise)
        t = C.c_int64()
        self._check(self._lib.vad_step_submit(self._h, _ptr(s, C.c_int64), s.size, T, f.ctypes.data_as(C.c_void_p), fmt, thr,
                                              C.byref(t)))
        self._tickets[int(t.value)] = (s, f, T, f0.ndim == 3)      # keeps the buffers alive until collected
        return int(t.value)

    def collect(self, ticket: int):
        """-> (probs, events, seg_frames) of a submitted ticket; blocks until its results are on the host."""
        held = self._tickets.get(int(ticket))
        if held is None:
            raise AudioProcessingError(f"Model prediction failed: ticket {ticket} is not outstanding")
        s, _f, T, multi = held
        probs = np.empty((s.size, T), np.float32)
        ev = np.zeros((s.size, T), np.uint8)
        seg = np.zeros(s.size, np.int32)
        try:
            self._check(self._lib.vad_step_collect(self._h, int(ticket), _ptr(probs, C.c_float), _ptr(ev, C.c_uint8),
                                                   _ptr(seg, C.c_int32)))
        finally:
            self._tickets.pop(int(ticket), None)
        return (probs, ev, seg) if multi else (probs[:, 0], ev[:, 0], seg)

    def step_rates(self, segments, slots, denoise: Optional[float] = 0.01):
        """One tick for streams at other input rates (``vad_step_rates``): ``segments`` = [(chunks [n_k, n_in_k] float32,
        sr_in_k), ...] with n_in = 256 / 512 / 768 / 1536 at 8 / 16 / 24 / 48 kHz; ``slots`` lists the streams of all
        segments in order.  Resample + model step chained on the GPU -> (probs, events, seg_frames)."""
        k = len(segments)
        arrs = [np.ascontiguousarray(a, np.float32) for a, _ in segments]
        for a in arrs:
            if a.ndim != 2:
                raise AudioProcessingError(f"Failed to resample audio: expected [n, n_in], got {a.shape}")
        ptrs = (C.c_void_p * k)(*[a.ctypes.data for a in arrs])
        n = (C.c_int64 * k)(*[a.shape[0] for a in arrs])
        sr = (C.c_int32 * k)(*[int(r) for _, r in segments])
        for a, (_, r) in zip(arrs, segments):
            want = {8000: 256, 16000: 512, 24000: 768, 48000: 1536}.get(int(r))
            if want is not None and a.shape[1] != want:
                raise AudioProcessingError(f"Failed to resample audio from {r}Hz to 16000Hz: a chunk must hold {want} samples, got {a.shape[1]}")
        s = np.ascontiguousarray(slots, dtype=np.int64).reshape(-1)
        total = sum(a.shape[0] for a in arrs)
        if s.size != total:
            raise AudioProcessingError(f"Model prediction failed: {s.size} slots for {total} chunks")
        probs = np.empty(total, np.float32)
        ev = np.zeros(total, np.uint8)
        seg = np.zeros(total, np.int32)
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step_rates(self._h, k, ptrs, n, sr, _ptr(s, C.c_int64), thr, _ptr(probs, C.c_float),
                                             _ptr(ev, C.c_uint8), _ptr(seg, C.c_int32)))
        return probs, ev, seg

    def step_rates_device(self, segments, d_probs: int, d_slots: int = 0, d_events: int = 0, d_seg: int = 0,
                          denoise: Optional[float] = 0.01, stream: int = 0) -> None:
        """Device-pointer form: ``segments`` = [(d_in, n, sr_in), ...] (``vad_step_rates_device``), asynchronous."""
        k = len(segments)
        ptrs = (C.c_void_p * k)(*[int(a[0]) for a in segments])
        n = (C.c_int64 * k)(*[int(a[1]) for a in segments])
        sr = (C.c_int32 * k)(*[int(a[2]) for a in segments])
        thr = -1.0 if denoise is None else float(denoise)
        self._check(self._lib.vad_step_rates_device(self._h, k, ptrs, n, sr, d_slots or None, thr, d_probs, d_events or None,
                                                    d_seg or None, stream or None))

    def resample_multi_device(self, segments, stream: int = 0) -> None:
        """One launch for up to 4 segments ``(d_in, n, n_in, sr_in, d_out)`` of device pointers (integers)."""
        k = len(segments)
        d_in = (C.c_void_p * k)(*[int(s[0]) for s in segments])
        n = (C.c_int64 * k)(*[int(s[1]) for s in segments])
        n_in = (C.c_int32 * k)(*[int(s[2]) for s in segments])
        sr = (C.c_int32 * k)(*[int(s[3]) for s in segments])
        d_out = (C.c_void_p * k)(*[int(s[4]) for s in segments])
        self._check(self._lib.vad_resample_multi_device(self._h, k, d_in, n, n_in, sr, d_out, stream or None))

    # ------------------------------------------------------------------ resampler (a11)
    def resample(self, chunks, sr_in: int) -> np.ndarray:
        """chunks [n, n_in] float32 at ``sr_in`` -> [n, 512] at 16 kHz (``vad_resample``)."""
        x = np.ascontiguousarray(chunks, np.float32)
        if x.ndim != 2:
            raise AudioProcessingError(f"Failed to resample audio: expected [n, n_in], got {x.shape}")
        out = np.empty((x.shape[0], 512), np.float32)
        self._check(self._lib.vad_resample(self._h, _ptr(x, C.c_float), x.shape[0], x.shape[1], int(sr_in),
                                           _ptr(out, C.c_float)))
        return out

    def resample_generic_device(self, d_in: int, rows: int, n_in: int, n_out: int, d_out: int, f64: bool = False) -> None:
        """Device-pointer form (``vad_resample_generic_device``): ``d_in`` -> [rows, n_in] float32 (float64 with ``f64``),
        ``d_out`` -> [rows, n_out] float32; synchronous - the result is complete on return."""
        self._check(self._lib.vad_resample_generic_device(self._h, C.c_void_p(int(d_in)), int(bool(f64)), int(rows), int(n_in), int(n_out),
                                                          C.c_void_p(int(d_out))))

    def set_resample_path(self, mode: int) -> None:
        """0 = by size, 1 = the direct kernel, 2 = the chirp-z / FFT path (``vad_debug_resample_path``)"""
        self._check(self._lib.vad_debug_resample_path(self._h, int(mode)), VADError)

    def resample_generic(self, arrays, n_out: int) -> np.ndarray:
        """arrays [rows, n_in] (float32, or float64 for double-precision input) -> [rows, n_out] float32: each row through
        ``scipy.signal.resample(row, n_out)`` as a whole (``vad_resample_generic``; any lengths)."""
        x = np.ascontiguousarray(arrays)
        if x.dtype != np.float64:
            x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 2:
            raise AudioProcessingError(f"Failed to resample audio: expected [rows, n_in], got {x.shape}")
        out = np.empty((x.shape[0], max(int(n_out), 0)), np.float32)
        self._check(self._lib.vad_resample_generic(self._h, x.ctypes.data_as(C.c_void_p), int(x.dtype == np.float64), x.shape[0],
                                                   x.shape[1], int(n_out), _ptr(out, C.c_float)))
        return out

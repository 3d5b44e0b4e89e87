"""ORACLE / TEST INFRASTRUCTURE — ctypes binding for oracle/silero_oracle.c.

Only tests/, tools/make_goldens.py, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  Product code under ``cutter_vad_amd/`` never
does (tests/test_boundary.py greps for it).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

EV_START, EV_END, EV_CONTINUE = 1, 2, 4


def build(force: bool = False) -> None:
    """Compile both oracle libraries with gcc (recipe: oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _lib(acc: str = "f64") -> C.CDLL:
    if acc in _LIBS:
        return _LIBS[acc]
    path = os.path.join(_HERE, f"liboracle_{acc}.so")
    src = os.path.join(_HERE, "silero_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    lib = C.CDLL(path)
    f32p = C.POINTER(C.c_float)
    lib.svo_load.restype = C.c_void_p
    lib.svo_load.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.svo_free.argtypes = [C.c_void_p]
    lib.svo_version.argtypes = [C.c_void_p]
    lib.svo_frame_samples.argtypes = [C.c_void_p]
    lib.svo_step.argtypes = [C.c_void_p, f32p, f32p, f32p]
    lib.svo_step_batch.argtypes = [C.c_void_p, f32p, C.c_int, f32p, f32p, C.c_int]
    lib.svo_denoise.argtypes = [f32p, C.c_int, C.c_float]
    lib.svo_pad_frame.argtypes = [f32p, C.c_int, f32p]
    lib.svo_num_frames.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.svo_split_frames.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
    lib.svo_resample.argtypes = [f32p, C.c_int, f32p, C.c_int]
    lib.svo_sm_sizeof.restype = C.c_size_t
    lib.svo_sm_init.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.svo_sm_reset.argtypes = [C.c_void_p]
    lib.svo_sm_step.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_longlong)]
    lib.svo_sm_active.argtypes = [C.c_void_p]
    lib.svo_sm_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_longlong)]
    lib.svo_wav_size.restype = C.c_size_t
    lib.svo_wav_size.argtypes = [C.c_longlong]
    lib.svo_write_wav16.restype = C.c_size_t
    lib.svo_write_wav16.argtypes = [f32p, C.c_longlong, C.c_int, C.c_char_p]
    _LIBS[acc] = lib
    return lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleModel:
    """Per-frame Silero model restatement.  ``acc='f64'`` (checker) or ``'f32'`` (CPU timing)."""

    STATE_FLOATS = 256

    def __init__(self, blob: bytes, acc: str = "f64"):
        self._lib = _lib(acc)
        err = C.create_string_buffer(256)
        self._h = self._lib.svo_load(blob, len(blob), err, len(err))
        if not self._h:
            raise ValueError(err.value.decode())
        self.version = self._lib.svo_version(self._h)
        self.frame_samples = self._lib.svo_frame_samples(self._h)      # 512; 256 for V5's 8 kHz sub-model

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.svo_free(h)

    def step(self, frame: np.ndarray, state: np.ndarray) -> float:
        """frame [512] f32, state [256] f32 (updated in place) -> probability."""
        assert frame.dtype == np.float32 and frame.shape == (self.frame_samples,) and frame.flags.c_contiguous
        assert state.dtype == np.float32 and state.shape == (256,) and state.flags.c_contiguous
        p = C.c_float()
        self._lib.svo_step(self._h, _fp(frame), _fp(state), C.byref(p))
        return float(p.value)

    def step_batch(self, frames: np.ndarray, states: np.ndarray, nthreads: int = 1) -> np.ndarray:
        """frames [n,512], states [n,256] (in place) -> probs [n]."""
        n = frames.shape[0]
        assert frames.dtype == np.float32 and frames.shape == (n, self.frame_samples) and frames.flags.c_contiguous
        assert states.dtype == np.float32 and states.shape == (n, 256) and states.flags.c_contiguous
        probs = np.empty(n, np.float32)
        self._lib.svo_step_batch(self._h, _fp(frames), n, _fp(states), _fp(probs), nthreads)
        return probs

    def run_stream(self, frames: np.ndarray, state: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """frames [T,512] of ONE stream, sequential -> (probs [T], final state [256])."""
        st = np.zeros(256, np.float32) if state is None else state.astype(np.float32).copy()
        out = np.empty(len(frames), np.float32)
        for t in range(len(frames)):
            out[t] = self.step(np.ascontiguousarray(frames[t], np.float32), st)
        return out, st


def denoise(x: np.ndarray, thr: float = 0.01) -> np.ndarray:
    y = np.ascontiguousarray(x, np.float32).copy()
    _lib().svo_denoise(_fp(y), y.size, thr)
    return y


def pad_frame(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(512, np.float32)
    _lib().svo_pad_frame(_fp(x), x.size, _fp(out))
    return out


def num_frames(n: int, frame: int, hop: int) -> int:
    return _lib().svo_num_frames(n, frame, hop)


def split_frames(x: np.ndarray, frame: int, hop: int) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    n = num_frames(x.size, frame, hop)
    if n < 0:
        raise ValueError("negative dimensions are not allowed")
    out = np.empty((n, frame), np.float32)
    if n:
        _lib().svo_split_frames(_fp(x), x.size, frame, hop, _fp(out))
    return out


def resample(x: np.ndarray, n_out: int) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(n_out, np.float32)
    _lib().svo_resample(_fp(x), x.size, _fp(out), n_out)
    return out


class StateMachine:
    """a10 restatement; one instance == one VADProcessor's hysteresis state."""

    def __init__(self, start_prob=0.7, end_prob=0.7, start_ratio=0.8, end_ratio=0.95, start_count=10, end_count=50):
        self._lib = _lib()
        self._buf = C.create_string_buffer(self._lib.svo_sm_sizeof())
        self._lib.svo_sm_init(self._buf, start_prob, end_prob, start_ratio, end_ratio, start_count, end_count)

    def reset(self) -> None:
        self._lib.svo_sm_reset(self._buf)

    def step(self, p: float, frame_len: int = 512) -> Tuple[int, int]:
        """-> (event bits, finished-segment length in samples or 0)."""
        seg = C.c_longlong()
        ev = self._lib.svo_sm_step(self._buf, float(p), frame_len, C.byref(seg))
        return ev, int(seg.value)

    @property
    def active(self) -> bool:
        return bool(self._lib.svo_sm_active(self._buf))

    def counts(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        act = self._lib.svo_sm_counts(self._buf, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(active=bool(act), n_start=a.value, n_end=b.value, buffered=c.value, seg_samples=d.value)


def wav16(x: np.ndarray, sample_rate: int = 16000) -> bytes:
    x = np.ascontiguousarray(x, np.float32)
    lib = _lib()
    buf = C.create_string_buffer(lib.svo_wav_size(x.size))
    n = lib.svo_write_wav16(_fp(x), x.size, sample_rate, buf)
    return buf.raw[:n]

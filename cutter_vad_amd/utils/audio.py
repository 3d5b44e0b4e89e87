"""Host-side audio helpers on the hot path's boundary.

Only the ``AudioUtils`` members that the reference calls on (or right next to) the per-frame
path are restated (/root/reference/src/real_time_vad/utils/audio.py): validation, mono mixdown,
framing, the denoise gate as applied to the audio KEPT for segments, PCM conversions, and
``resample_audio``.  The offline helpers (normalise / window / RMS / file I/O) are out of scope
(SURVEY §2 row 6).

On the hot path itself the gate and the int16 scaling are fused into the HIP kernel's load
(csrc/silero_v5.hip); ``resample_audio`` is a HIP kernel behind ``vad_resample`` — none of the
model arithmetic ever runs in this file.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from ..core.exceptions import AudioProcessingError

_CHUNK_IN = {8000: 256, 24000: 768, 48000: 1536}


class AudioUtils:
    @staticmethod
    def validate_audio_data(audio_data: np.ndarray) -> None:
        """audio.py:211-231"""
        if not isinstance(audio_data, np.ndarray):
            raise AudioProcessingError("Audio data must be a numpy array")
        if audio_data.size == 0:
            raise AudioProcessingError("Audio data is empty")
        if not np.isfinite(audio_data).all():
            raise AudioProcessingError("Audio data contains infinite or NaN values")
        if audio_data.ndim > 2:
            raise AudioProcessingError(f"Audio data has too many dimensions: {audio_data.ndim}")

    @staticmethod
    def convert_to_mono(audio_data: np.ndarray) -> np.ndarray:
        """audio.py:193-208: mean over axis 1 for [N, C] input."""
        if audio_data.ndim == 1:
            return audio_data
        if audio_data.ndim == 2:
            return np.mean(audio_data, axis=1)
        raise AudioProcessingError(f"Unsupported audio shape: {audio_data.shape}")

    @staticmethod
    def split_into_frames(audio_data: np.ndarray, frame_size: int, hop_size: Optional[int] = None) -> np.ndarray:
        """audio.py:164-190: ``(len - frame) // hop + 1`` frames, tail dropped, no carry-over.
        A chunk shorter than ``frame - hop`` makes the count negative and numpy raises, as in
        the reference."""
        hop = frame_size // 2 if hop_size is None else hop_size
        count = (len(audio_data) - frame_size) // hop + 1
        frames = np.zeros((count, frame_size), dtype=audio_data.dtype)      # count < 0 raises here, like the reference
        if 0 < count <= 32:                       # the usual chunk: one to a few frames, plain row copies are fastest
            for i in range(count):
                frames[i] = audio_data[i * hop:i * hop + frame_size]
        elif count > 0:
            frames[:] = np.lib.stride_tricks.sliding_window_view(audio_data, frame_size)[::hop][:count]
        return frames

    @staticmethod
    def denoise_audio(audio_data: np.ndarray, noise_threshold: float = 0.01) -> np.ndarray:
        """audio.py:104-121: strict ``|x| > threshold`` gate.  Used on the host only for the
        audio that is stored into voice segments; the model sees the same gate inside the kernel."""
        try:
            return np.where(np.abs(audio_data) > noise_threshold, audio_data, 0.0)
        except Exception as e:  # pragma: no cover
            raise AudioProcessingError(f"Failed to denoise audio: {e}")

    @staticmethod
    def pcm_to_float32(pcm_data: bytes, bit_depth: int = 16) -> np.ndarray:
        """audio.py:294-316 (note: /32768, while the websocket server divides by 32767)."""
        try:
            if bit_depth == 16:
                return np.frombuffer(pcm_data, dtype=np.int16).astype(np.float32) / 32768.0
            if bit_depth == 32:
                return np.frombuffer(pcm_data, dtype=np.int32).astype(np.float32) / 2147483648.0
            raise ValueError(f"Unsupported bit depth: {bit_depth}")
        except Exception as e:
            raise AudioProcessingError(f"Failed to convert PCM to float32: {e}")

    @staticmethod
    def float32_to_pcm(audio_data: np.ndarray, bit_depth: int = 16) -> bytes:
        """audio.py:318-341"""
        try:
            if bit_depth == 16:
                return (audio_data * 32767).astype(np.int16).tobytes()
            if bit_depth == 32:
                return (audio_data * 2147483647).astype(np.int32).tobytes()
            raise ValueError(f"Unsupported bit depth: {bit_depth}")
        except Exception as e:
            raise AudioProcessingError(f"Failed to convert float32 to PCM: {e}")

    @staticmethod
    def resample_audio(audio_data: np.ndarray, original_rate: int, target_rate: int) -> np.ndarray:
        """audio.py:19-55: ``scipy.signal.resample(audio_data, int(len(audio_data) * target_rate / original_rate))`` of the
        WHOLE array (Fourier method) as float32 - on the GPU, for every input the reference accepts: any length, any pair of
        rates, [N] or [N, C] (axis 0), float / integer / bool dtypes (complex input: the reference keeps the real part of the
        complex resample, which is the resample of the real part).

        A 1-D float32 array that is exactly one streaming chunk (256 / 768 / 1536 samples at 8 / 24 / 48 kHz -> 512 at
        16 kHz) takes the MFMA kernel the serving tick uses (``vad_resample``); everything else ``vad_resample_generic``:
        small calls evaluate the Fourier operator entry by entry in float64 (never stored), from 2^25 entries up the same
        function runs as two chirp-z transforms on power-of-two float64 FFTs (O(n log n), up to 2^25 samples).  All three
        are the same function of the input up to float32 rounding (tests: <= 1e-5 against scipy; measured <= 4e-7).  An
        array beyond both kernels' limits is refused with AudioProcessingError - never cut into pieces, which would give a
        different answer from the reference's."""
        try:
            if original_rate == target_rate:
                return audio_data
            ratio = target_rate / original_rate
            resampled_length = int(len(audio_data) * ratio)
            return _fourier_resample(np.asarray(audio_data), resampled_length)
        except Exception as e:
            if isinstance(e, AudioProcessingError) and "Failed to resample audio" in str(e):
                raise
            raise AudioProcessingError(
                f"Failed to resample audio from {original_rate}Hz to {target_rate}Hz: {e}",
                f"Input shape: {getattr(audio_data, 'shape', None)}, dtype: {getattr(audio_data, 'dtype', None)}")


def _fourier_resample(x: np.ndarray, num: int) -> np.ndarray:
    """``scipy.signal.resample(x, num).astype(np.float32)`` along axis 0, window=None (audio.py:46-49)."""
    if num < 0:
        raise ValueError("negative dimensions are not allowed")                # numpy's, from scipy's np.zeros(newshape)
    n = x.shape[0]
    if n == 0:
        raise ValueError("invalid number of data points (0) specified")        # scipy.fft's, from rfft of an empty axis
    if np.iscomplexobj(x):
        x = x.real
    tail = x.shape[1:]
    # scipy transforms float32 (and float16) input in single precision, everything else real in double: the kernel takes both
    wide = x.dtype not in (np.float32, np.float16)
    rows = np.array(np.moveaxis(x, 0, -1).reshape(-1, n), np.float64 if wide else np.float32, order="C")     # [columns, n], a copy
    # a NaN / Inf anywhere reaches every output sample of its column through the transform
    bad = ~np.isfinite(rows).all(axis=1)
    rows[bad] = 0
    if num == 0:
        y = np.zeros((rows.shape[0], 1), np.float32)          # scipy's irfft hands back ONE sample per column, times num / len = 0
    elif x.ndim == 1 and not wide and num == 512 and n in _CHUNK_IN.values():
        from ..pool import default_pool
        rate = next(sr for sr, k in _CHUNK_IN.items() if k == n)
        y = default_pool().any_engine().resample(rows, rate)
    else:
        from ..pool import default_pool
        y = default_pool().any_engine().resample_generic(rows, num)
    y[bad] = np.nan
    return np.ascontiguousarray(np.moveaxis(y.reshape(tail + (y.shape[1],)), -1, 0))

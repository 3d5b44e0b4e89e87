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
        """audio.py:19-55 (``scipy.signal.resample``, Fourier method), on the GPU.

        Chunking convention of this build (the reference defines none, it never calls this on
        its path — vad_wrapper.py:621-624 is ``pass``): the signal is resampled in independent
        chunks that each yield 512 output samples at 16 kHz (256 / 768 / 1536 input samples at
        8 / 24 / 48 kHz).  A whole-array call with exactly one chunk is therefore identical to
        the reference; other lengths must be a whole number of chunks."""
        if original_rate == target_rate:
            return audio_data
        try:
            if target_rate != 16000 or original_rate not in _CHUNK_IN:
                raise ValueError("the HIP resampler converts 8/24/48 kHz to 16 kHz")
            n_in = _CHUNK_IN[original_rate]
            x = np.ascontiguousarray(audio_data, dtype=np.float32)
            if x.ndim != 1 or x.size == 0 or x.size % n_in:
                raise ValueError(f"length must be a positive multiple of {n_in} samples at {original_rate} Hz")
            from ..pool import default_pool
            return default_pool().resample(x.reshape(-1, n_in), original_rate).reshape(-1)
        except AudioProcessingError:
            raise
        except Exception as e:
            raise AudioProcessingError(
                f"Failed to resample audio from {original_rate}Hz to {target_rate}Hz: {e}",
                f"Input shape: {getattr(audio_data, 'shape', None)}, dtype: {getattr(audio_data, 'dtype', None)}")

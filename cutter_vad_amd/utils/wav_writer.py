"""In-memory RIFF/WAVE payload for finished voice segments.

Runs once per segment end, not per frame.  Byte-compatible with the reference's
``WAVWriter`` (/root/reference/src/real_time_vad/utils/wav_writer.py:40-136): 44-byte PCM header,
``clip(x * 32767, -32768, 32767)`` truncated to int16 (or the 32-bit analogue)."""

from __future__ import annotations

import struct

import numpy as np

from ..core.exceptions import AudioProcessingError

_SCALE = {16: (32767, -32768, 32767, np.int16), 32: (2147483647, -2147483648, 2147483647, np.int32)}


class WAVWriter:
    def __init__(self, sample_rate: int = 16000, bit_depth: int = 16, channels: int = 1) -> None:
        if bit_depth not in _SCALE:
            raise ValueError(f"Unsupported bit depth: {bit_depth}. Must be 16 or 32.")
        if channels not in (1, 2):
            raise ValueError(f"Unsupported channel count: {channels}. Must be 1 or 2.")
        self.sample_rate, self.bit_depth, self.channels = sample_rate, bit_depth, channels

    def header(self, data_size: int) -> bytes:
        bps = self.bit_depth // 8
        return (b"RIFF" + struct.pack("<I", 36 + data_size) + b"WAVE" + b"fmt "
                + struct.pack("<IHHIIHH", 16, 1, self.channels, self.sample_rate,
                              self.sample_rate * self.channels * bps, self.channels * bps, self.bit_depth)
                + b"data" + struct.pack("<I", data_size))

    def write_wav_data(self, audio_data: np.ndarray) -> bytes:
        try:
            if not isinstance(audio_data, np.ndarray):
                raise ValueError("Audio data must be a numpy array")
            x = audio_data if audio_data.dtype == np.float32 else audio_data.astype(np.float32)
            if self.channels == 1 and x.ndim > 1:
                x = np.mean(x, axis=1)
            mul, lo, hi, dt = _SCALE[self.bit_depth]
            pcm = np.clip(x * mul, lo, hi).astype(dt).tobytes()
            return self.header(len(pcm)) + pcm
        except Exception as e:
            raise AudioProcessingError(f"Failed to create WAV data: {e}")

    def write_wav_file(self, filename: str, audio_data: np.ndarray) -> None:
        try:
            with open(filename, "wb") as f:
                f.write(self.write_wav_data(audio_data))
        except AudioProcessingError:
            raise
        except Exception as e:
            raise AudioProcessingError(f"Failed to write WAV file {filename}: {e}")

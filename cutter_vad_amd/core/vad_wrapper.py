"""``VADWrapper`` — the public drop-in surface.

Public methods, argument meaning, defaults, locking and error wrapping follow the reference
(/root/reference/src/real_time_vad/core/vad_wrapper.py:202-984) so that code written against
``real_time_vad.VADWrapper`` runs unchanged; the per-frame work happens in the HIP engine slot
owned by the wrapper's ``VADProcessor``.  Differences that are deliberate:
  * all wrappers of a process share one engine per (model, GPU) instead of one ORT session each;
  * per-frame INFO/DEBUG logging in ``_handle_callbacks`` (vad_wrapper.py:494-514) is dropped.
"""

from __future__ import annotations

import threading
import time
import warnings
from contextlib import contextmanager
from typing import Any, Callable, ClassVar, Dict, List, Optional, Union

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, ValidationError, field_validator, model_validator

from ..utils.audio import AudioUtils
from .config import SampleRate, SileroModelVersion, VADConfig
from .exceptions import AudioProcessingError, CallbackError, ConfigurationError, VADError
from .silero_model import ProcessingResult, ProcessingStatistics, VADProcessor

VoiceStartCallback = Callable[[], None]
VoiceEndCallback = Callable[[bytes], None]
VoiceContinueCallback = Callable[[bytes], None]


class VADWrapperState(BaseModel):
    """vad_wrapper.py:25-82"""
    model_config = ConfigDict(arbitrary_types_allowed=True, validate_assignment=True, extra="forbid")
    is_initialized: bool = False
    total_frames_processed: int = Field(default=0, ge=0)
    total_processing_time: float = Field(default=0.0, ge=0.0)
    last_error: Optional[str] = None

    @property
    def average_processing_time_per_frame(self) -> float:
        return self.total_processing_time / self.total_frames_processed if self.total_frames_processed else 0.0

    def reset_statistics(self) -> None:
        self.total_frames_processed = 0
        self.total_processing_time = 0.0

    def record_error(self, error: Exception) -> None:
        self.last_error = str(error)

    def clear_error(self) -> None:
        self.last_error = None


class CallbackConfiguration(BaseModel):
    """vad_wrapper.py:85-127"""
    model_config = ConfigDict(arbitrary_types_allowed=True, validate_assignment=True, extra="forbid")
    voice_start_callback: Optional[VoiceStartCallback] = None
    voice_end_callback: Optional[VoiceEndCallback] = None
    voice_continue_callback: Optional[VoiceContinueCallback] = None

    @field_validator("voice_start_callback", "voice_end_callback", "voice_continue_callback")
    @classmethod
    def _callable(cls, v):
        if v is not None and not callable(v):
            raise ValueError("Callback must be a callable function")
        return v

    def has_any_callback(self) -> bool:
        return any(cb is not None for cb in (self.voice_start_callback, self.voice_end_callback,
                                             self.voice_continue_callback))


class ThresholdConfiguration(BaseModel):
    """vad_wrapper.py:130-199 — note the end-count default 57 here vs 50 in VADConfig."""
    model_config = ConfigDict(validate_assignment=True, extra="forbid")
    vad_start_probability: float = Field(default=0.7, ge=0.0, le=1.0)
    vad_end_probability: float = Field(default=0.7, ge=0.0, le=1.0)
    voice_start_ratio: float = Field(default=0.8, ge=0.0, le=1.0)
    voice_end_ratio: float = Field(default=0.95, ge=0.0, le=1.0)
    voice_start_frame_count: int = Field(default=10, ge=1)
    voice_end_frame_count: int = Field(default=57, ge=1)

    @model_validator(mode="after")
    def _sane(self):
        if self.vad_start_probability < 0.1:
            raise ValueError("Start probability should be at least 0.1 for reliable detection")
        if self.vad_end_probability < 0.1:
            raise ValueError("End probability should be at least 0.1 for reliable detection")
        if self.voice_start_frame_count > 100:
            raise ValueError("Voice start frame count should not exceed 100 for responsive detection")
        if self.voice_end_frame_count > 200:
            raise ValueError("Voice end frame count should not exceed 200 for responsive detection")
        return self


class VADWrapper:
    _DEFAULT_FRAME_OVERLAP_RATIO: ClassVar[float] = 0.5
    _MAX_PROCESSING_TIME_WARNING: ClassVar[float] = 1.0

    def __init__(self, config: Optional[VADConfig] = None) -> None:
        try:
            self._config = config if config is not None else VADConfig()
            if not isinstance(self._config, VADConfig):
                raise ValueError("Config must be a VADConfig instance")
            if self._config.buffer_size <= 0:
                raise ValueError("Buffer size must be positive")
            self._state = VADWrapperState()
            self._callbacks = CallbackConfiguration()
            self._lock = threading.Lock()
            self._processor: Optional[VADProcessor] = None
            self._initialize_processor()
        except ValidationError as e:
            raise VADError(f"Invalid configuration provided: {e}")
        except Exception as e:
            raise VADError(f"Failed to initialize VAD wrapper: {e}")

    def _initialize_processor(self) -> None:
        try:
            old = self._processor
            self._processor = self._make_processor(self._config)
            if old is not None:
                old.close()
            self._state.is_initialized = True
            self._state.clear_error()
        except Exception as e:
            self._state.record_error(e)
            self._state.is_initialized = False
            raise VADError(f"Failed to initialize VAD processor: {e}")

    @staticmethod
    def _make_processor(config: VADConfig) -> VADProcessor:
        return VADProcessor(config)

    # ========================= configuration =========================

    def set_sample_rate(self, sample_rate: SampleRate) -> None:
        with self._lock:
            try:
                if not isinstance(sample_rate, SampleRate):
                    raise ValueError(f"Invalid sample rate type: {type(sample_rate)}")
                old = self._config.sample_rate
                self._config.sample_rate = sample_rate
                if old != sample_rate and self._state.is_initialized:
                    self._initialize_processor()
            except ValidationError as e:
                raise ConfigurationError("sample_rate", str(sample_rate), str(e))
            except Exception as e:
                self._state.record_error(e)
                raise ConfigurationError("sample_rate", str(sample_rate), str(e))

    def set_silero_model(self, model_version: SileroModelVersion) -> None:
        with self._lock:
            try:
                if not isinstance(model_version, SileroModelVersion):
                    raise ValueError(f"Invalid model version type: {type(model_version)}")
                old = self._config.model_version
                self._config.model_version = model_version
                if old != model_version and self._state.is_initialized:
                    self._initialize_processor()
            except ValidationError as e:
                raise ConfigurationError("model_version", str(model_version), str(e))
            except Exception as e:
                self._state.record_error(e)
                raise ConfigurationError("model_version", str(model_version), str(e))

    def set_thresholds(self, vad_start_probability: float = 0.7, vad_end_probability: float = 0.7,
                       voice_start_ratio: float = 0.8, voice_end_ratio: float = 0.95,
                       voice_start_frame_count: int = 10, voice_end_frame_count: int = 57) -> None:
        with self._lock:
            try:
                t = ThresholdConfiguration(
                    vad_start_probability=vad_start_probability, vad_end_probability=vad_end_probability,
                    voice_start_ratio=voice_start_ratio, voice_end_ratio=voice_end_ratio,
                    voice_start_frame_count=voice_start_frame_count, voice_end_frame_count=voice_end_frame_count)
                for name in ThresholdConfiguration.model_fields:
                    setattr(self._config, name, getattr(t, name))
                if self._processor:
                    self._processor.reset()
            except ValidationError as e:
                raise ConfigurationError("thresholds", "multiple", str(e))
            except Exception as e:
                self._state.record_error(e)
                raise ConfigurationError("thresholds", "multiple", str(e))

    # ========================= callbacks =========================

    def set_callbacks(self, voice_start_callback: Optional[VoiceStartCallback] = None,
                      voice_end_callback: Optional[VoiceEndCallback] = None,
                      voice_continue_callback: Optional[VoiceContinueCallback] = None) -> None:
        try:
            self._callbacks = CallbackConfiguration(voice_start_callback=voice_start_callback,
                                                    voice_end_callback=voice_end_callback,
                                                    voice_continue_callback=voice_continue_callback)
        except ValidationError as e:
            raise VADError(f"Invalid callback configuration: {e}")

    def _execute_callback_safely(self, callback: Optional[Callable], callback_name: str, *args, **kwargs) -> None:
        if callback is None:
            return
        try:
            callback(*args, **kwargs)
        except Exception as e:
            self._state.record_error(e)
            raise CallbackError(callback_name, e)

    def _handle_callbacks(self, result: ProcessingResult) -> None:
        """vad_wrapper.py:478-522: START -> cb(); END only with wav bytes; CONTINUE only with pcm bytes."""
        try:
            if not isinstance(result, ProcessingResult):
                raise ValueError("Invalid processing result type")
            if result.voice_started:
                self._execute_callback_safely(self._callbacks.voice_start_callback, "voice_start")
            if result.voice_ended and result.wav_data:
                self._execute_callback_safely(self._callbacks.voice_end_callback, "voice_end", result.wav_data)
            if result.voice_continuing and result.pcm_data:
                self._execute_callback_safely(self._callbacks.voice_continue_callback, "voice_continue", result.pcm_data)
        except ValidationError as e:
            raise CallbackError("result_validation", e)

    # ========================= audio =========================

    @contextmanager
    def _processing_context(self):
        if not self._state.is_initialized or self._processor is None:
            raise VADError("VAD processor not initialized")
        t0 = time.time()
        try:
            yield
        finally:
            dt = time.time() - t0
            self._state.total_processing_time += dt
            if dt > self._MAX_PROCESSING_TIME_WARNING:
                warnings.warn(f"Audio processing took {dt:.3f}s, which may indicate performance issues")

    def process_audio_data(self, audio_data: Union[np.ndarray, List[float]]) -> None:
        with self._lock:
            with self._processing_context():
                try:
                    self._process_audio_frames(self._validate_and_prepare_audio(audio_data))
                except ValidationError as e:
                    self._state.record_error(e)
                    raise AudioProcessingError(f"Audio validation failed: {e}")
                except Exception as e:
                    self._state.record_error(e)
                    raise AudioProcessingError(f"Audio processing failed: {e}")

    def _validate_and_prepare_audio(self, audio_data: Union[np.ndarray, List[float]]) -> np.ndarray:
        if isinstance(audio_data, list):
            if not audio_data:
                raise AudioProcessingError("Audio data cannot be empty")
            audio = np.array(audio_data, dtype=np.float32)
        elif isinstance(audio_data, np.ndarray):
            audio = audio_data.astype(np.float32)
        else:
            raise AudioProcessingError(f"Unsupported audio data type: {type(audio_data)}")
        AudioUtils.validate_audio_data(audio)
        return AudioUtils.convert_to_mono(audio)

    def _process_audio_frames(self, audio_data: np.ndarray) -> None:
        """vad_wrapper.py:610-647: frames of ``buffer_size`` at hop ``buffer_size // 2``, no carry-over; a callback
        that raises aborts the remaining frames of this call (their state is NOT advanced).  The frames of a chunk
        go to the engine in one launch (``VADProcessor.process_frames`` keeps the abort contract)."""
        try:
            frame_size = self._config.buffer_size
            hop = int(frame_size * self._DEFAULT_FRAME_OVERLAP_RATIO)
            frames = [np.pad(f, (0, frame_size - len(f))) if len(f) < frame_size else f
                      for f in AudioUtils.split_into_frames(audio_data, frame_size, hop)]
            if not frames:
                return
            batched = getattr(self._processor, "process_frames", None)
            if batched is None or len(frames) == 1:
                results = (self._processor.process_frame(f) for f in frames)
            else:
                results = batched(np.stack(frames))
            try:
                for result in results:
                    self._handle_callbacks(result)
                    self._state.total_frames_processed += 1
            finally:
                results.close()
        except Exception as e:
            raise AudioProcessingError(f"Frame processing failed: {e}")

    def process_audio_data_with_buffer(self, audio_buffer: np.ndarray, count: int) -> None:
        try:
            if not isinstance(audio_buffer, np.ndarray):
                raise AudioProcessingError("Audio buffer must be a numpy array")
            if count < 0:
                raise AudioProcessingError("Count must be non-negative")
            if count > len(audio_buffer):
                raise AudioProcessingError(f"Count {count} exceeds buffer size {len(audio_buffer)}")
            self.process_audio_data(audio_buffer[:count])
        except Exception as e:
            if not isinstance(e, (AudioProcessingError, ValidationError)):
                raise AudioProcessingError(f"Buffer processing failed: {e}")
            raise

    # ========================= state / info =========================

    @property
    def processor(self) -> Optional[VADProcessor]:
        return self._processor

    @property
    def config(self) -> VADConfig:
        return self._config

    @config.setter
    def config(self, value: VADConfig) -> None:
        self.update_config(value)

    def reset(self) -> None:
        with self._lock:
            try:
                if self._processor:
                    self._processor.reset()
                self._state.reset_statistics()
                self._state.clear_error()
            except Exception as e:
                self._state.record_error(e)
                raise VADError(f"Failed to reset VAD state: {e}")

    def cleanup(self) -> None:
        with self._lock:
            try:
                if self._processor is not None:
                    self._processor.close()   # hands the engine slot back to the pool
                self._processor = None
                self._state.is_initialized = False
                self._state.clear_error()
            except Exception as e:
                self._state.record_error(e)

    def get_statistics(self) -> Dict[str, Any]:
        with self._lock:
            try:
                stats = {
                    "total_frames_processed": self._state.total_frames_processed,
                    "total_processing_time": self._state.total_processing_time,
                    "average_processing_time_per_frame": self._state.average_processing_time_per_frame,
                    "is_initialized": self._state.is_initialized,
                    "last_error": self._state.last_error,
                    "has_callbacks": self._callbacks.has_any_callback(),
                    "config": self._serialize_config_for_json(),
                }
                if self._processor:
                    ps = self._processor.get_statistics()
                    stats.update(ps.model_dump() if isinstance(ps, ProcessingStatistics) else ps)
                return stats
            except Exception as e:
                self._state.record_error(e)
                return {"error": str(e), "is_initialized": self._state.is_initialized,
                        "total_frames_processed": self._state.total_frames_processed}

    def _serialize_config_for_json(self) -> Dict[str, Any]:
        try:
            d = self._config.model_dump()
            mv = d.get("model_version")
            if mv is not None:
                d["model_version"] = mv.value if hasattr(mv, "value") else str(mv)
            if d.get("model_path") is not None:
                d["model_path"] = str(d["model_path"])
            return d
        except Exception as e:
            return {"sample_rate": int(self._config.sample_rate), "model_version": self._config.model_version.value,
                    "error": f"Serialization error: {e}"}

    def get_config(self) -> VADConfig:
        return self._config

    def update_config(self, config: VADConfig) -> None:
        with self._lock:
            old_config = self._config
            try:
                if not isinstance(config, VADConfig):
                    raise ValueError("Config must be a VADConfig instance")
                config.model_validate(config.model_dump())
                if self._processor:
                    self._processor.update_config(config)
                else:
                    self._config = config
                    self._initialize_processor()
                self._config = config
                self._state.clear_error()
            except ValidationError as e:
                self._state.record_error(e)
                raise VADError(f"Invalid configuration: {e}")
            except Exception as e:
                self._config = old_config
                self._state.record_error(e)
                raise VADError(f"Failed to update configuration: {e}")

    def is_voice_active(self) -> bool:
        try:
            return bool(self._processor.is_voice_active) if self._processor else False
        except Exception as e:
            self._state.record_error(e)
            return False

    def get_last_error(self) -> Optional[str]:
        return self._state.last_error

    def get_last_error_details(self) -> Dict[str, Any]:
        return {"last_error": self._state.last_error, "is_initialized": self._state.is_initialized,
                "total_frames_processed": self._state.total_frames_processed,
                "has_processor": self._processor is not None}

    # ========================= context / repr =========================

    def __enter__(self) -> "VADWrapper":
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        try:
            self.cleanup()
        except Exception:
            pass

    def __del__(self) -> None:
        try:
            self.cleanup()
        except Exception:
            pass

    def __repr__(self) -> str:
        return (f"VADWrapper(initialized={self._state.is_initialized}, sample_rate={self._config.sample_rate}, "
                f"model_version={self._config.model_version}, frames_processed={self._state.total_frames_processed})")

    def __str__(self) -> str:
        status = "Initialized" if self._state.is_initialized else "Not Initialized"
        return (f"VAD Wrapper - {status}\nSample Rate: {self._config.sample_rate.value} Hz\n"
                f"Model Version: {self._config.model_version.value}\n"
                f"Frames Processed: {self._state.total_frames_processed}\n"
                f"Has Callbacks: {self._callbacks.has_any_callback()}")

"""``VADWrapper`` — the public drop-in surface over one slot of the shared HIP engine.

Same public methods, argument meaning, defaults, locking rule and exception classes / message prefixes as the
reference's ``VADWrapper`` (/root/reference/src/real_time_vad/core/vad_wrapper.py:202-984), so that code written against
``real_time_vad.VADWrapper`` runs unchanged.  The implementation is this package's own:

  * configuration changes go through two tables (``_ENUM_SETTERS``, ``_THRESHOLD_RULES``) and one error-mapping helper
    (``_mapped``) instead of a hand-written try/except per method;
  * callbacks are a name -> callable table dispatched from ``_EVENTS``;
  * counters are a plain dataclass;
  * all wrappers of a process share one engine per (model, GPU) - a wrapper owns a slot, not an inference session - and a
    chunk's frames go to the engine in one launch (``VADProcessor.process_frames``).
The reference's per-frame INFO / DEBUG logging (vad_wrapper.py:494-514) is not reproduced.
"""

from __future__ import annotations

import threading
import time
import warnings
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple, Type, Union

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, ValidationError, create_model, model_validator

from ..utils.audio import AudioUtils
from .config import SampleRate, SileroModelVersion, VADConfig
from .exceptions import AudioProcessingError, CallbackError, ConfigurationError, VADError
from .silero_model import ProcessingResult, ProcessingStatistics, VADProcessor

VoiceStartCallback = Callable[[], None]
VoiceEndCallback = Callable[[bytes], None]
VoiceContinueCallback = Callable[[bytes], None]

# set_thresholds: (VADConfig field, default of the call, schema bounds, "reasonable" bound checked after the schema: which side,
# value, message) - vad_wrapper.py:130-199, 367-419.  The end-count default is 57 here and 50 in VADConfig: the reference's quirk,
# kept.
_THRESHOLD_RULES: Tuple[Tuple[str, Any, Dict[str, Any], Optional[Tuple[str, float, str]]], ...] = (
    ("vad_start_probability", 0.7, dict(ge=0.0, le=1.0), ("min", 0.1, "Start probability should be at least 0.1 for reliable detection")),
    ("vad_end_probability", 0.7, dict(ge=0.0, le=1.0), ("min", 0.1, "End probability should be at least 0.1 for reliable detection")),
    ("voice_start_ratio", 0.8, dict(ge=0.0, le=1.0), None),
    ("voice_end_ratio", 0.95, dict(ge=0.0, le=1.0), None),
    ("voice_start_frame_count", 10, dict(ge=1), ("max", 100, "Voice start frame count should not exceed 100 for responsive detection")),
    ("voice_end_frame_count", 57, dict(ge=1), ("max", 200, "Voice end frame count should not exceed 200 for responsive detection")),
)


def _reasonable(model):
    """the reference's after-validator (vad_wrapper.py:183-199): the two probabilities first, then the two counts"""
    for name, _default, _bounds, rule in _THRESHOLD_RULES:
        if rule is not None:
            side, limit, message = rule
            v = getattr(model, name)
            if (side == "min" and v < limit) or (side == "max" and v > limit):
                raise ValueError(message)
    return model


# The validation model set_thresholds goes through, like the reference's: pydantic in lax mode, so numpy scalars, numeric
# strings, bools and integral floats for the counts are coerced exactly as the reference coerces them, and a value outside a
# bound fails with pydantic's own message for THAT bound.  Built from the rules table above.
ThresholdConfiguration = create_model(
    "ThresholdConfiguration", __config__=ConfigDict(validate_assignment=True, extra="forbid"),
    __validators__={"validate_threshold_logic": model_validator(mode="after")(_reasonable)},
    **{name: (type(default), Field(default=default, **bounds)) for name, default, bounds, _rule in _THRESHOLD_RULES})


class CallbackConfiguration(BaseModel):
    """Import compatibility with the reference module (vad_wrapper.py:84-127); the wrapper itself keeps its callbacks in a
    name -> callable table."""
    model_config = ConfigDict(arbitrary_types_allowed=True, validate_assignment=True, extra="forbid")
    voice_start_callback: Optional[Callable[[], None]] = None
    voice_end_callback: Optional[Callable[[bytes], None]] = None
    voice_continue_callback: Optional[Callable[[bytes], None]] = None

    def has_any_callback(self) -> bool:
        return any(getattr(self, k) is not None for k in type(self).model_fields)


class VADWrapperState(BaseModel):
    """Import compatibility with the reference module (vad_wrapper.py:27-82): a snapshot type with the reference's field names;
    ``VADWrapper.state_snapshot()`` fills one from the wrapper's counters."""
    model_config = ConfigDict(validate_assignment=True, extra="forbid")
    is_initialized: bool = False
    total_frames_processed: int = Field(default=0, ge=0)
    total_processing_time: float = Field(default=0.0, ge=0.0)
    last_error: Optional[str] = None

    @property
    def average_processing_time_per_frame(self) -> float:
        return self.total_processing_time / self.total_frames_processed if self.total_frames_processed else 0.0

    def reset_statistics(self) -> None:
        self.total_frames_processed, self.total_processing_time = 0, 0.0

    def record_error(self, error: Exception) -> None:
        self.last_error = str(error)

    def clear_error(self) -> None:
        self.last_error = None


# set_sample_rate / set_silero_model: config field -> (enum it must be an instance of, what the error calls it)
_ENUM_SETTERS: Dict[str, Tuple[type, str]] = {
    "sample_rate": (SampleRate, "sample rate"),
    "model_version": (SileroModelVersion, "model version"),
}
# ProcessingResult flag -> (payload attribute or None, callback key): delivery order on a frame is start, end, continue
# (vad_wrapper.py:478-522); END / CONTINUE fire only when their payload is non-empty
_EVENTS: Tuple[Tuple[str, Optional[str], str], ...] = (
    ("voice_started", None, "voice_start"),
    ("voice_ended", "wav_data", "voice_end"),
    ("voice_continuing", "pcm_data", "voice_continue"),
)
_FRAME_HOP_RATIO = 0.5            # vad_wrapper.py:238, 628: frames overlap by half
_SLOW_CALL_SECONDS = 1.0          # vad_wrapper.py:239: warn when one call takes longer


@dataclass
class _Counters:
    initialized: bool = False
    frames: int = 0
    seconds: float = 0.0
    last_error: Optional[str] = None

    @property
    def seconds_per_frame(self) -> float:
        return self.seconds / self.frames if self.frames else 0.0


class VADWrapper:
    def __init__(self, config: Optional[VADConfig] = None) -> None:
        self._lock = threading.Lock()             # one call at a time per wrapper, callbacks included (vad_wrapper.py:560)
        self._n = _Counters()
        self._cb: Dict[str, Optional[Callable]] = {key: None for _, _, key in _EVENTS}
        self._processor: Optional[VADProcessor] = None
        try:
            self._config = VADConfig() if config is None else config
            if not isinstance(self._config, VADConfig):
                raise ValueError("Config must be a VADConfig instance")
            if self._config.buffer_size <= 0:
                raise ValueError("Buffer size must be positive")
            self._build_processor()
        except ValidationError as e:
            raise VADError(f"Invalid configuration provided: {e}")
        except Exception as e:
            raise VADError(f"Failed to initialize VAD wrapper: {e}")

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _make_processor(config: VADConfig) -> VADProcessor:
        return VADProcessor(config)

    def _build_processor(self) -> None:
        """A fresh processor (= a fresh engine slot) for the current configuration; the old slot goes back to the pool."""
        try:
            fresh = self._make_processor(self._config)
        except Exception as e:
            self._n.last_error, self._n.initialized = str(e), False
            raise VADError(f"Failed to initialize VAD processor: {e}")
        old, self._processor = self._processor, fresh
        if old is not None:
            old.close()
        self._n.initialized, self._n.last_error = True, None

    @contextmanager
    def _mapped(self, make: Callable[[Exception], Exception], validation: Optional[Callable[[Exception], Exception]] = None,
                passthrough: Tuple[Type[BaseException], ...] = ()) -> Iterator[None]:
        """Run a block; whatever it raises is recorded as the last error and re-raised as ``make(e)`` (pydantic validation
        errors as ``validation(e)`` when given).  This is the wrapper's one error-wrapping rule."""
        try:
            yield
        except passthrough:
            raise
        except ValidationError as e:
            self._n.last_error = str(e)
            raise (validation or make)(e)
        except Exception as e:
            self._n.last_error = str(e)
            raise make(e)

    # ------------------------------------------------------------------ configuration
    def _set_enum(self, field: str, value: Any) -> None:
        kind, label = _ENUM_SETTERS[field]
        with self._lock, self._mapped(lambda e: ConfigurationError(field, str(value), str(e))):
            if not isinstance(value, kind):
                raise ValueError(f"Invalid {label} type: {type(value)}")
            changed = getattr(self._config, field) != value
            setattr(self._config, field, value)
            if changed and self._n.initialized:
                self._build_processor()           # the reference rebuilds its session here (vad_wrapper.py:326-327, 358-359)

    def set_sample_rate(self, sample_rate: SampleRate) -> None:
        self._set_enum("sample_rate", sample_rate)

    def set_silero_model(self, model_version: SileroModelVersion) -> None:
        self._set_enum("model_version", model_version)

    def set_thresholds(self, vad_start_probability: float = 0.7, vad_end_probability: float = 0.7,
                       voice_start_ratio: float = 0.8, voice_end_ratio: float = 0.95,
                       voice_start_frame_count: int = 10, voice_end_frame_count: int = 57) -> None:
        given = dict(zip((r[0] for r in _THRESHOLD_RULES), (vad_start_probability, vad_end_probability, voice_start_ratio,
                                                           voice_end_ratio, voice_start_frame_count, voice_end_frame_count)))
        wrap = lambda e: ConfigurationError("thresholds", "multiple", str(e))       # noqa: E731
        with self._lock, self._mapped(wrap, wrap):
            checked = ThresholdConfiguration(**given)           # coerces and validates like the reference (pydantic, lax mode)
            for name in given:
                setattr(self._config, name, getattr(checked, name))    # VADConfig validates on assignment as well
            if self._processor:
                self._processor.reset()           # counters restart under the new thresholds (vad_wrapper.py:412-413)

    def state_snapshot(self) -> VADWrapperState:
        return VADWrapperState(is_initialized=self._n.initialized, total_frames_processed=self._n.frames,
                               total_processing_time=self._n.seconds, last_error=self._n.last_error)

    # ------------------------------------------------------------------ callbacks
    def set_callbacks(self, voice_start_callback: Optional[VoiceStartCallback] = None,
                      voice_end_callback: Optional[VoiceEndCallback] = None,
                      voice_continue_callback: Optional[VoiceContinueCallback] = None) -> None:
        table = {"voice_start": voice_start_callback, "voice_end": voice_end_callback, "voice_continue": voice_continue_callback}
        for name, fn in table.items():
            if fn is not None and not callable(fn):
                raise VADError(f"Invalid callback configuration: {name}_callback: Callback must be a callable function")
        self._cb = table

    def _deliver(self, result: ProcessingResult) -> None:
        if not isinstance(result, ProcessingResult):
            raise CallbackError("result_validation", ValueError("Invalid processing result type"))
        for flag, payload, key in _EVENTS:
            if not getattr(result, flag):
                continue
            args = ()
            if payload is not None:
                data = getattr(result, payload)
                if not data:
                    continue
                args = (data,)
            fn = self._cb[key]
            if fn is None:
                continue
            try:
                fn(*args)
            except Exception as e:                # a raising callback aborts the rest of the chunk (vad_wrapper.py:474-476, 646-647)
                self._n.last_error = str(e)
                raise CallbackError(key, e)

    # ------------------------------------------------------------------ audio
    def process_audio_data(self, audio_data: Union[np.ndarray, List[float]]) -> None:
        with self._lock:
            if not self._n.initialized or self._processor is None:
                raise VADError("VAD processor not initialized")
            t0 = time.time()
            try:
                with self._mapped(lambda e: AudioProcessingError(f"Audio processing failed: {e}"),
                                  lambda e: AudioProcessingError(f"Audio validation failed: {e}")):
                    self._run_chunk(self._as_mono_float(audio_data))
            finally:
                dt = time.time() - t0
                self._n.seconds += dt
                if dt > _SLOW_CALL_SECONDS:
                    warnings.warn(f"Audio processing took {dt:.3f}s, which may indicate performance issues")

    @staticmethod
    def _as_mono_float(audio_data: Union[np.ndarray, List[float]]) -> np.ndarray:
        """list / ndarray -> float32 copy, finite / ndim / size checks, stereo -> mono (vad_wrapper.py:576-608)"""
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

    def _run_chunk(self, audio: np.ndarray) -> None:
        """Frames of ``buffer_size`` at hop ``buffer_size // 2``, no carry-over between calls, tail dropped
        (vad_wrapper.py:610-647).  The chunk's frames advance in ONE launch; ``process_frames`` steps the stream back if a
        callback stops the loop early, so later frames count as unprocessed exactly as in the reference."""
        try:
            size = self._config.buffer_size
            frames = AudioUtils.split_into_frames(audio, size, int(size * _FRAME_HOP_RATIO))
            if len(frames) == 0:
                return
            batched = getattr(self._processor, "process_frames", None)
            results = (batched(np.asarray(frames)) if batched is not None and len(frames) > 1
                       else (self._processor.process_frame(f) for f in frames))
            try:
                for result in results:
                    self._deliver(result)
                    self._n.frames += 1
            finally:
                results.close()
        except Exception as e:
            raise AudioProcessingError(f"Frame processing failed: {e}")

    def process_audio_data_with_buffer(self, audio_buffer: np.ndarray, count: int) -> None:
        with self._mapped(lambda e: AudioProcessingError(f"Buffer processing failed: {e}"),
                          passthrough=(AudioProcessingError, ValidationError)):
            if not isinstance(audio_buffer, np.ndarray):
                raise AudioProcessingError("Audio buffer must be a numpy array")
            if count < 0:
                raise AudioProcessingError("Count must be non-negative")
            if count > len(audio_buffer):
                raise AudioProcessingError(f"Count {count} exceeds buffer size {len(audio_buffer)}")
            self.process_audio_data(audio_buffer[:count])

    # ------------------------------------------------------------------ state / info
    @property
    def processor(self) -> Optional[VADProcessor]:
        return self._processor

    @property
    def config(self) -> VADConfig:
        return self._config

    @config.setter
    def config(self, value: VADConfig) -> None:
        self.update_config(value)

    def get_config(self) -> VADConfig:
        return self._config

    def update_config(self, config: VADConfig) -> None:
        with self._lock:
            previous = self._config
            try:
                with self._mapped(lambda e: VADError(f"Failed to update configuration: {e}"),
                                  lambda e: VADError(f"Invalid configuration: {e}")):
                    if not isinstance(config, VADConfig):
                        raise ValueError("Config must be a VADConfig instance")
                    config.model_validate(config.model_dump())
                    if self._processor:
                        self._processor.update_config(config)
                        self._config = config
                    else:
                        self._config = config
                        self._build_processor()
                    self._n.last_error = None
            except VADError:
                self._config = previous           # a failed update leaves the wrapper on its previous configuration
                raise

    def reset(self) -> None:
        with self._lock, self._mapped(lambda e: VADError(f"Failed to reset VAD state: {e}")):
            if self._processor:
                self._processor.reset()
            self._n.frames, self._n.seconds, self._n.last_error = 0, 0.0, None

    def cleanup(self) -> None:
        with self._lock:
            try:
                if self._processor is not None:
                    self._processor.close()       # hands the engine slot back to the pool
                self._processor = None
                self._n.initialized, self._n.last_error = False, None
            except Exception as e:
                self._n.last_error = str(e)

    def is_voice_active(self) -> bool:
        try:
            return bool(self._processor.is_voice_active) if self._processor else False
        except Exception as e:
            self._n.last_error = str(e)
            return False

    def get_last_error(self) -> Optional[str]:
        return self._n.last_error

    def get_last_error_details(self) -> Dict[str, Any]:
        return {"last_error": self._n.last_error, "is_initialized": self._n.initialized,
                "total_frames_processed": self._n.frames, "has_processor": self._processor is not None}

    def get_statistics(self) -> Dict[str, Any]:
        with self._lock:
            base = {"total_frames_processed": self._n.frames, "is_initialized": self._n.initialized}
            try:
                stats = dict(base, total_processing_time=self._n.seconds,
                             average_processing_time_per_frame=self._n.seconds_per_frame, last_error=self._n.last_error,
                             has_callbacks=any(fn is not None for fn in self._cb.values()), config=self._config_as_json())
                if self._processor:
                    ps = self._processor.get_statistics()
                    stats.update(ps.model_dump() if isinstance(ps, ProcessingStatistics) else ps)
                return stats
            except Exception as e:
                self._n.last_error = str(e)
                return dict(base, error=str(e))

    def _config_as_json(self) -> Dict[str, Any]:
        try:
            d = self._config.model_dump()
            d["model_version"] = getattr(d.get("model_version"), "value", d.get("model_version"))
            if d.get("model_path") is not None:
                d["model_path"] = str(d["model_path"])
            return d
        except Exception as e:
            return {"sample_rate": int(self._config.sample_rate), "model_version": self._config.model_version.value,
                    "error": f"Serialization error: {e}"}

    # ------------------------------------------------------------------ context / repr
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
        return (f"VADWrapper(initialized={self._n.initialized}, sample_rate={self._config.sample_rate}, "
                f"model_version={self._config.model_version}, frames_processed={self._n.frames})")

    def __str__(self) -> str:
        return "\n".join((f"VAD Wrapper - {'Initialized' if self._n.initialized else 'Not Initialized'}",
                          f"Sample Rate: {self._config.sample_rate.value} Hz",
                          f"Model Version: {self._config.model_version.value}",
                          f"Frames Processed: {self._n.frames}",
                          f"Has Callbacks: {any(fn is not None for fn in self._cb.values())}"))

"""Per-stream model operator and processor over the shared HIP engine.

Same names, constructor signatures, return types and error behaviour as the reference's
``SileroVADModel`` / ``VADProcessor`` (/root/reference/src/real_time_vad/core/silero_model.py),
but nothing here computes: ``predict`` / ``process_frame`` forward one frame to the engine slot
this object owns (C ABI ``vad_step`` / ``vad_step_events``), where the denoise gate, the model and
the hysteresis state machine run fused on the GPU.  The host keeps only what callbacks need:
the audio of the current segment (pre-roll + frames) and the counters ``get_statistics`` reports.
"""

from __future__ import annotations

import logging
import os
from collections import deque
from pathlib import Path
from typing import Any, Deque, Dict, List, Optional

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, field_validator, model_validator

from .. import _ffi, weights_io
from ..pool import _VERSION_INT, EnginePool, default_pool, resolve_model_path
from ..utils.audio import AudioUtils
from ..utils.wav_writer import WAVWriter
from .config import SileroModelVersion, VADConfig
from .exceptions import AudioProcessingError, ModelInitializationError, ModelNotFoundError

__all__ = ["ModelState", "ProcessingResult", "ProcessingStatistics", "ModelConfiguration", "SileroVADModel",
           "VADProcessor"]


# ========================= data models (same fields as the reference) =========================

class ModelState(BaseModel):
    """LSTM state as ONNX lays it out (silero_model.py:33-83): V5 ``state`` (2,1,128); V4
    ``hidden_state`` / ``cell_state`` (2,1,64).  A snapshot of the engine slot."""
    model_config = ConfigDict(arbitrary_types_allowed=True, validate_assignment=True, extra="forbid")
    state: Optional[np.ndarray] = None
    hidden_state: Optional[np.ndarray] = None
    cell_state: Optional[np.ndarray] = None

    @field_validator("state", "hidden_state", "cell_state")
    @classmethod
    def _f32(cls, v):
        if v is not None:
            if not isinstance(v, np.ndarray):
                raise ValueError("State must be a numpy array")
            if v.dtype != np.float32:
                raise ValueError("State arrays must be float32")
        return v

    @model_validator(mode="after")
    def _exclusive(self):
        if self.state is not None and (self.hidden_state is not None or self.cell_state is not None):
            raise ValueError("Cannot have both combined state and separate states")
        return self


class ProcessingResult(BaseModel):
    """silero_model.py:86-138"""
    model_config = ConfigDict(arbitrary_types_allowed=True, validate_assignment=True, extra="forbid")
    voice_started: bool = False
    voice_ended: bool = False
    voice_continuing: bool = False
    probability: float = Field(ge=0.0, le=1.0)
    wav_data: Optional[bytes] = None
    pcm_data: Optional[bytes] = None


class ProcessingStatistics(BaseModel):
    """silero_model.py:141-199"""
    model_config = ConfigDict(arbitrary_types_allowed=True, validate_assignment=True, extra="forbid")
    is_voice_active: bool
    voice_start_frame_count: int = Field(ge=0)
    voice_end_frame_count: int = Field(ge=0)
    recent_probabilities: List[float] = Field(default_factory=list)
    average_probability: float = Field(ge=0.0, le=1.0)
    voice_buffer_size: int = Field(ge=0)
    current_voice_length: int = Field(ge=0)

    @field_validator("recent_probabilities")
    @classmethod
    def _in_range(cls, v: List[float]) -> List[float]:
        for p in v:                                                     # silero_model.py:191-199
            if not 0.0 <= p <= 1.0:
                raise ValueError(f"Probability {p} must be between 0.0 and 1.0")
        return v


class ModelConfiguration(BaseModel):
    """silero_model.py:202-232.  Besides ``.onnx`` an ``.svw`` weight blob is accepted."""
    model_config = ConfigDict(validate_assignment=True, extra="forbid")
    model_path: str
    model_version: SileroModelVersion

    @field_validator("model_path")
    @classmethod
    def _exists(cls, v: str) -> str:
        if not os.path.exists(v):
            raise ModelNotFoundError(v)
        if not os.path.isfile(v):
            raise ValueError(f"Model path must be a file: {v}")
        if not v.endswith((".onnx", ".svw")):
            raise ValueError(f"Model file must have .onnx extension: {v}")
        return v


class _Session:
    """What the reference exposes as ``model.session`` (an ORT ``InferenceSession``): here the
    shared engine; ``get_providers`` keeps ``get_model_info`` working."""

    def __init__(self, engine):
        self.engine = engine

    def get_providers(self) -> List[str]:
        return ["HIPFusedSileroProvider"]


# ========================= the operator =========================

class SileroVADModel:
    """``SileroVADModel(model_path, model_version)`` — one stream's view of the engine."""

    def __init__(self, model_path: str, model_version: SileroModelVersion, *, pool: Optional[EnginePool] = None,
                 device_id: Optional[int] = None) -> None:
        self.config = ModelConfiguration(model_path=model_path, model_version=model_version)
        self.prediction_count = 0
        self.session: Optional[_Session] = None
        self._slot: Optional[int] = None
        self._pool = pool or default_pool()
        self._device_id = device_id
        # the reference's session holds both sub-models of the graph and picks one per call from ``sr``; here each is
        # an engine of its own: {is_8k: (session, slot)}, the 8 kHz one (V4 only) is created on first use
        self._variants: Dict[bool, tuple] = {}
        self._k8 = False
        self._load_model()
        self._reset_states()

    # -- silero_model.py:303-334
    def _load_model(self, sample_rate: int = 16000) -> None:
        try:
            k8 = weights_io.is_8k_variant(_VERSION_INT[self.config.model_version], sample_rate)
            eng = self._pool.engine_for(self.config.model_path, self.config.model_version, self._device_id,
                                        sample_rate=sample_rate)
            self._variants[k8] = (_Session(eng), eng.open_stream())
            self.session, self._slot = self._variants[k8]
            self._k8 = k8
        except (ModelNotFoundError, ModelInitializationError):
            raise
        except Exception as e:
            raise ModelInitializationError(self.config.model_version.value,
                                           f"Failed to load model from {self.config.model_path}: {e}")

    def frame_samples(self, sample_rate: int) -> int:
        """Samples one model step takes at this `sr`: 512, or 256 on Silero V5's 8 kHz sub-model."""
        return weights_io.frame_samples(_VERSION_INT[self.config.model_version], sample_rate)

    def select_rate(self, sample_rate: int, frame_len: int = 512) -> None:
        """What ``Equal(sr, 16000)`` does inside the graph: choose the sub-model for this call.  The recurrent state
        is the graph's ``h`` / ``c`` inputs, shared by both branches, so it moves with the switch - and so does the rest
        of the stream (thresholds, counters and histories of the device state machine, which belong to the processor,
        not to a sub-model): the whole slot travels as one ``vad_stream_save`` / ``vad_stream_restore`` blob."""
        self._check_rate(sample_rate, self.config.model_version, frame_len)
        k8 = weights_io.is_8k_variant(_VERSION_INT[self.config.model_version], sample_rate)
        if k8 == self._k8:
            return
        stream = self.engine.save_stream(self._slot)
        if k8 in self._variants:
            self.session, self._slot = self._variants[k8]
            self._k8 = k8
        else:
            self._load_model(sample_rate)
        self.engine.restore_stream(self._slot, stream)

    @property
    def engine(self):
        if self.session is None:
            raise ModelInitializationError(self.config.model_version.value, "Model not loaded")
        return self.session.engine

    @property
    def slot(self) -> int:
        return self._slot

    # -- silero_model.py:384-401
    def _reset_states(self) -> None:
        for sess, slot in self._variants.values():
            sess.engine.reset([slot])

    @property
    def model_state(self) -> ModelState:
        hc = self.engine.get_state(self._slot)
        if self.config.model_version == SileroModelVersion.V5:
            return ModelState(state=hc.reshape(2, 1, 128).copy())
        return ModelState(hidden_state=hc[:128].reshape(2, 1, 64).copy(), cell_state=hc[128:].reshape(2, 1, 64).copy())

    # -- silero_model.py:403-447
    def predict(self, audio_chunk: np.ndarray, sample_rate: int) -> float:
        try:
            if self.session is None:
                raise ModelInitializationError(self.config.model_version.value, "Model not loaded")
            self.select_rate(sample_rate, len(audio_chunk))
            frame = self._prepare_audio_input(audio_chunk, self.frame_samples(sample_rate))
            p = float(self.engine.step([self._slot], frame, denoise=None)[0])
            p = self._extract_probability(p)
            self.prediction_count += 1
            return p
        except (ModelInitializationError, AudioProcessingError):
            raise
        except Exception as e:
            raise AudioProcessingError(f"Model prediction failed: {e}")

    @staticmethod
    def _check_rate(sample_rate: int, model_version: SileroModelVersion = SileroModelVersion.V5, frame_len: int = 512) -> None:
        # The graphs select the 16 kHz weights only for sr == 16000.  V4's other branch (its 8 kHz sub-model, taken for
        # 8 / 24 / 48 kHz alike) takes the same 512-sample frames.  V5's is built for native 8 kHz audio in 256-sample frames
        # (VADConfig(sample_rate=8000, buffer_size=256)): that is served; with the reference's 512-sample frames - or at
        # 24 / 48 kHz - a 3-D tensor reaches its LSTM-cell subgraph and onnxruntime refuses (reproduced by
        # oracle/onnx_interp.py), so those raise here as they do there.
        if int(sample_rate) != 16000 and model_version == SileroModelVersion.V5:
            if int(sample_rate) == 8000 and frame_len <= 256:
                return
            raise AudioProcessingError(
                f"Model prediction failed: sample rate {sample_rate} selects the 8 kHz graph branch, which takes native "
                f"8 kHz audio in 256-sample frames (got sr = {sample_rate}, {frame_len} samples); resample to 16 kHz first")

    # -- silero_model.py:449-474
    @staticmethod
    def _prepare_audio_input(audio_chunk: np.ndarray, frame_samples: int = 512) -> np.ndarray:
        try:
            n, L = len(audio_chunk), frame_samples
            if n != L:
                audio_chunk = np.pad(audio_chunk, (0, L - n)) if n < L else audio_chunk[:L]
            return np.ascontiguousarray(audio_chunk, dtype=np.float32).reshape(1, -1)
        except Exception as e:
            raise AudioProcessingError(f"Audio input preparation failed: {e}")

    # -- silero_model.py:501-524
    @staticmethod
    def _extract_probability(p: float) -> float:
        try:
            if not (0.0 <= p <= 1.0):
                raise ValueError(f"Invalid probability value: {p}")
            return p
        except Exception as e:
            raise AudioProcessingError(f"Probability extraction failed: {e}")

    def reset(self) -> None:
        self._reset_states()

    def close(self) -> None:
        for sess, slot in self._variants.values():
            try:
                sess.engine.close_stream(slot)
            except Exception:
                pass
        self._variants = {}
        self._slot = None
        self.session = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- silero_model.py:548-566
    def get_model_info(self) -> Dict[str, Any]:
        st = self.model_state if self.session else ModelState()
        prov = self.session.get_providers() if self.session else None
        return {
            "model_path": self.config.model_path,
            "model_version": self.config.model_version.value,
            "prediction_count": self.prediction_count,
            "session_providers": prov,
            "has_cuda": "CUDAExecutionProvider" in (prov or []),
            "state_shape": {
                "state": st.state.shape if st.state is not None else None,
                "hidden_state": st.hidden_state.shape if st.hidden_state is not None else None,
                "cell_state": st.cell_state.shape if st.cell_state is not None else None,
            },
        }


# ========================= segment bookkeeping =========================

class SegmentAssembler:
    """Host half of VADProcessor._process_voice_state (silero_model.py:790-949): the audio
    buffers and counters.  The start/end DECISIONS come from the device state machine (event
    bits of ``vad_step_events``); this class replays the same arithmetic for bookkeeping and
    refuses to continue if the two ever disagree."""

    def __init__(self, config: VADConfig, wav_writer: WAVWriter) -> None:
        self.config = config
        self.wav_writer = wav_writer
        self.clear()

    def clear(self) -> None:
        self.is_voice_active = False
        self.voice_start_frame_count = 0
        self.voice_end_frame_count = 0
        self.recent_start_frames: Deque[bool] = deque(maxlen=20)
        self.recent_end_frames: Deque[bool] = deque(maxlen=100)
        self.voice_buffer: Deque[np.ndarray] = deque()
        self.current_voice_data: Optional[np.ndarray] = None

    def push(self, probability: float, audio_frame: np.ndarray, device_events: Optional[int]) -> Dict[str, Any]:
        c = self.config
        res: Dict[str, Any] = dict(voice_started=False, voice_ended=False, voice_continuing=False, wav_data=None,
                                   pcm_data=None)
        if not self.is_voice_active:
            above = probability >= c.vad_start_probability
            self.recent_start_frames.append(above)
            if above:
                self.voice_start_frame_count += 1
                self.voice_buffer.append(audio_frame.copy())
                n = c.voice_start_frame_count
                if self.voice_start_frame_count >= n and len(self.recent_start_frames) >= n:
                    window = list(self.recent_start_frames)[-n:]
                    if sum(window) / len(window) >= c.voice_start_ratio:
                        self.is_voice_active = True
                        self.voice_start_frame_count = 0
                        self.voice_end_frame_count = 0
                        if self.voice_buffer:
                            self.current_voice_data = np.concatenate(list(self.voice_buffer))
                        self.voice_buffer.clear()
                        res["voice_started"] = True
            else:
                self.voice_start_frame_count = 0
                self.voice_buffer.clear()
        else:
            self.current_voice_data = (audio_frame.copy() if self.current_voice_data is None
                                       else np.concatenate([self.current_voice_data, audio_frame]))
            res["voice_continuing"] = True
            res["pcm_data"] = audio_frame.tobytes()
            below = probability < c.vad_end_probability
            self.recent_end_frames.append(below)
            if below:
                self.voice_end_frame_count += 1
                n = c.voice_end_frame_count
                if self.voice_end_frame_count >= n and len(self.recent_end_frames) >= n:
                    window = list(self.recent_end_frames)[-n:]
                    if sum(window) / len(window) >= c.voice_end_ratio:
                        wav = None
                        if self.current_voice_data is not None:
                            wav = self.wav_writer.write_wav_data(self.current_voice_data)
                        self.is_voice_active = False
                        self.voice_end_frame_count = 0
                        self.current_voice_data = None
                        res["voice_ended"] = True
                        res["wav_data"] = wav
            else:
                self.voice_end_frame_count = 0
        if device_events is not None:
            host = ((_ffi.VAD_EV_START if res["voice_started"] else 0) | (_ffi.VAD_EV_END if res["voice_ended"] else 0)
                    | (_ffi.VAD_EV_CONTINUE if res["voice_continuing"] else 0))
            if host != int(device_events):
                raise AudioProcessingError(
                    f"state machine divergence: device events {int(device_events)} vs host bookkeeping {host}")
        return res


# ========================= the per-stream processor =========================

class VADProcessor:
    """silero_model.py:569-1033 — same public surface."""

    def __init__(self, config: VADConfig, *, pool: Optional[EnginePool] = None, device_id: Optional[int] = None) -> None:
        self.config = config
        self.model: Optional[SileroVADModel] = None
        self.voice_probabilities: Deque[float] = deque(maxlen=100)
        self.wav_writer = WAVWriter(sample_rate=config.output_wav_sample_rate, bit_depth=config.output_wav_bit_depth,
                                    channels=1)
        self._seg = SegmentAssembler(config, self.wav_writer)
        self._pool = pool
        self._device_id = device_id
        self._synced = None
        self._load_model()

    # state lives in the assembler; expose the reference's attribute names
    is_voice_active = property(lambda s: s._seg.is_voice_active)
    voice_start_frame_count = property(lambda s: s._seg.voice_start_frame_count)
    voice_end_frame_count = property(lambda s: s._seg.voice_end_frame_count)
    recent_start_frames = property(lambda s: s._seg.recent_start_frames)
    recent_end_frames = property(lambda s: s._seg.recent_end_frames)
    voice_buffer = property(lambda s: s._seg.voice_buffer)
    current_voice_data = property(lambda s: s._seg.current_voice_data)

    def _get_model_directory(self) -> Path:
        if self.config.model_path:
            logging.info(f"Using configured model path: {self.config.model_path}")
            return Path(self.config.model_path)
        return Path(resolve_model_path(self.config)).parent

    def _load_model(self) -> None:
        try:
            path = Path(resolve_model_path(self.config))
            if not path.exists():
                raise ModelNotFoundError(str(path))
            if self.model is not None:
                self.model.close()
            self.model = SileroVADModel(str(path), self.config.model_version, pool=self._pool,
                                        device_id=self._device_id)
            self._synced = None
        except (ModelNotFoundError, ModelInitializationError):
            raise
        except Exception as e:
            raise ModelInitializationError(self.config.model_version.value, f"Failed to initialize VAD processor: {e}")

    def _sync_thresholds(self) -> None:
        c = self.config
        cur = (c.vad_start_probability, c.vad_end_probability, c.voice_start_ratio, c.voice_end_ratio,
               c.voice_start_frame_count, c.voice_end_frame_count)
        key = (id(self.model.engine), self.model.slot, cur)      # per (engine, slot): a rate switch changes both
        if key != self._synced:
            self.model.engine.set_thresholds(self.model.slot, *cur)
            self._synced = key

    # -- silero_model.py:723-762
    def process_frame(self, audio_frame: np.ndarray) -> ProcessingResult:
        try:
            if self.model is None:
                raise ModelInitializationError(self.config.model_version.value, "Model not loaded")
            kept = self._preprocess_audio_frame(audio_frame)
            sr = int(self.config.sample_rate)
            self.model.select_rate(sr, len(audio_frame))             # first: the thresholds go to the slot that will run
            self._sync_thresholds()
            frame = SileroVADModel._prepare_audio_input(np.asarray(audio_frame), self.model.frame_samples(sr))
            thr = 0.01 if self.config.enable_denoising else None
            try:
                p, ev, _seg = self.model.engine.step_events([self.model.slot], frame, denoise=thr)
            except AudioProcessingError:
                raise
            except Exception as e:
                raise AudioProcessingError(f"Model prediction failed: {e}")
            probability = SileroVADModel._extract_probability(float(p[0]))
            self.model.prediction_count += 1
            self.voice_probabilities.append(probability)
            data = self._seg.push(probability, kept, int(ev[0]))
            data["probability"] = probability
            return ProcessingResult(**data)
        except (ModelInitializationError, AudioProcessingError):
            raise
        except Exception as e:
            raise AudioProcessingError(f"Frame processing failed: {e}")

    def process_frames(self, frames: np.ndarray):
        """The F frames of one chunk (``[F, 512]``, in order) in ONE launch (``vad_step_multi``): generator of the
        ``ProcessingResult`` of each frame, in order, with the host bookkeeping applied as each result is taken.

        The reference runs ``process_frame`` per frame (vad_wrapper.py:632-641) and a callback that raises leaves the
        remaining frames of the chunk unprocessed (:646-647).  To keep that contract the stream is saved before the
        launch; if the consumer stops after frame i (generator closed early) the stream is restored and frames
        0..i are replayed, so the device ends exactly where the reference's model and counters would be.
        A frame that fails validation raises when its turn comes, after the frames before it were delivered."""
        frames = np.asarray(frames)
        if frames.ndim != 2 or frames.shape[0] <= 1:
            for f in frames.reshape(-1, frames.shape[-1]) if frames.ndim == 2 else [frames]:
                yield self.process_frame(f)
            return
        if self.model is None:
            raise ModelInitializationError(self.config.model_version.value, "Model not loaded")
        kept, bad = [], None
        for f in frames:
            try:
                kept.append(self._preprocess_audio_frame(f))
            except AudioProcessingError as e:
                bad = AudioProcessingError(f"Frame processing failed: {e}") if "Frame processing" not in str(e) else e
                break
        F = len(kept)
        if F:
            try:
                sr = int(self.config.sample_rate)
                self.model.select_rate(sr, frames.shape[1])
                self._sync_thresholds()
                L = self.model.frame_samples(sr)
                x = np.stack([SileroVADModel._prepare_audio_input(np.asarray(f), L)[0] for f in frames[:F]])[None]
                thr = 0.01 if self.config.enable_denoising else None
                eng, slot = self.model.engine, self.model.slot
                saved = eng.save_stream(slot)
                try:
                    p, ev = eng.step_multi([slot], x, denoise=thr)
                except AudioProcessingError:
                    raise
                except Exception as e:
                    raise AudioProcessingError(f"Model prediction failed: {e}")
            except (ModelInitializationError, AudioProcessingError):
                raise
            except Exception as e:
                raise AudioProcessingError(f"Frame processing failed: {e}")
            done = 0
            try:
                for i in range(F):
                    try:
                        probability = SileroVADModel._extract_probability(float(p[0, i]))
                        self.model.prediction_count += 1
                        self.voice_probabilities.append(probability)
                        data = self._seg.push(probability, kept[i], int(ev[0, i]))
                        data["probability"] = probability
                        result = ProcessingResult(**data)
                    except (ModelInitializationError, AudioProcessingError):
                        raise
                    except Exception as e:
                        raise AudioProcessingError(f"Frame processing failed: {e}")
                    done = i + 1
                    yield result
            finally:
                if done < F:       # stopped early: put the device where the reference would be (after frame done-1)
                    eng.restore_stream(slot, saved)
                    if done:
                        eng.step_multi([slot], x[:, :done], denoise=thr)
        if bad is not None:
            raise bad

    # -- silero_model.py:764-788: the frame that is KEPT for segments (the model gates in-kernel)
    def _preprocess_audio_frame(self, audio_frame: np.ndarray) -> np.ndarray:
        try:
            AudioUtils.validate_audio_data(audio_frame)
            if self.config.enable_denoising:
                audio_frame = AudioUtils.denoise_audio(audio_frame)
            return audio_frame
        except Exception as e:
            raise AudioProcessingError(f"Audio preprocessing failed: {e}")

    # -- silero_model.py:951-968
    def reset(self) -> None:
        if self.model:
            self.model.reset()
            self._synced = None   # vad_stream_reset keeps thresholds; re-push in case config changed
        self._seg.clear()
        self.voice_probabilities.clear()

    def get_statistics(self) -> ProcessingStatistics:
        s = self._seg
        return ProcessingStatistics(
            is_voice_active=s.is_voice_active, voice_start_frame_count=s.voice_start_frame_count,
            voice_end_frame_count=s.voice_end_frame_count, recent_probabilities=list(self.voice_probabilities),
            average_probability=float(np.mean(self.voice_probabilities)) if self.voice_probabilities else 0.0,
            voice_buffer_size=len(s.voice_buffer),
            current_voice_length=len(s.current_voice_data) if s.current_voice_data is not None else 0)

    def get_model_info(self) -> Dict[str, Any]:
        if self.model:
            return self.model.get_model_info()
        return {"model_path": None, "model_version": self.config.model_version.value, "model_loaded": False}

    # -- silero_model.py:1002-1033
    def update_config(self, new_config: VADConfig) -> None:
        model_changed = (new_config.model_version != self.config.model_version
                         or new_config.model_path != self.config.model_path)
        self.config = new_config
        self._seg.config = new_config
        self.wav_writer = WAVWriter(sample_rate=new_config.output_wav_sample_rate,
                                    bit_depth=new_config.output_wav_bit_depth, channels=1)
        self._seg.wav_writer = self.wav_writer
        if model_changed:
            self._load_model()
        self.reset()

    def close(self) -> None:
        if self.model is not None:
            self.model.close()
            self.model = None

"""Process-wide engine registry and the batched multi-stream driver.

The reference builds one ORT session per ``VADWrapper``
(/root/reference/src/real_time_vad/core/silero_model.py:321-325, one wrapper per websocket
client: websocket_service/server/vad_websocket_server.py:277).  Here every wrapper of a
process that uses the same (model file, device) shares ONE :class:`~cutter_vad_amd.engine.Engine`
and owns one slot of its stream pool; :class:`StreamBatch` is the new multi-stream caller that
advances many slots per launch (SURVEY §7 "the pool's submit/step protocol is new design").

Engine-only knobs live here, not in ``VADConfig`` (which must keep ``extra="forbid"``):
    VAD_DEVICE_ID     HIP device ordinal (default: LOCAL_RANK, else 0)
    VAD_MAX_STREAMS   slots per engine   (default: 8192)
"""

from __future__ import annotations

import os
import threading
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import weights_io
from .core.config import SileroModelVersion, VADConfig
from .core.exceptions import ModelInitializationError, ModelNotFoundError
from .engine import Engine

_VERSION_INT = {SileroModelVersion.V4: 4, SileroModelVersion.V5: 5}


def default_device() -> int:
    return int(os.environ.get("VAD_DEVICE_ID", os.environ.get("LOCAL_RANK", "0")))


def default_max_streams() -> int:
    return int(os.environ.get("VAD_MAX_STREAMS", "8192"))


class EnginePool:
    """(resolved model file, device) -> shared Engine."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self._engines: Dict[Tuple[str, int, bool], Engine] = {}

    def engine_for(self, model_path: str, version: SileroModelVersion, device_id: Optional[int] = None,
                   max_streams: Optional[int] = None, sample_rate: int = 16000) -> Engine:
        """``sample_rate`` picks the sub-model the graph would run: V4 has a second set of weights for every rate
        but 16 000; it is a separate engine (own packed weights, same kernels + the 2-step LSTM tail)."""
        dev = default_device() if device_id is None else device_id
        k8 = weights_io.is_8k_variant(_VERSION_INT[version], sample_rate)
        if k8 and model_path.endswith("_16k.svw"):
            model_path = model_path[:-len("_16k.svw")] + "_8k.svw"
        key = (os.path.abspath(model_path), dev, k8)
        with self._lock:
            eng = self._engines.get(key)
            if eng is None or not eng.handle:
                if not os.path.exists(model_path):
                    raise ModelNotFoundError(model_path)
                try:
                    blob = weights_io.load_weight_blob(model_path, _VERSION_INT[version], sample_rate)
                except weights_io.WeightFormatError as e:
                    # same wording as the reference's signature check, silero_model.py:378-382
                    msg = str(e)
                    if msg.startswith("Expected"):
                        msg = f"Model signature validation failed: {msg}"
                    raise ModelInitializationError(version.value, f"Failed to load model from {model_path}: {msg}")
                eng = Engine(blob, model_version=_VERSION_INT[version], device_id=dev,
                             max_streams=max_streams or default_max_streams(), sample_rate=8000 if k8 else 16000)
                self._engines[key] = eng
            return eng

    def any_engine(self) -> Engine:
        with self._lock:
            for eng in self._engines.values():
                if eng.handle:
                    return eng
        return self.engine_for(resolve_model_path(VADConfig()), SileroModelVersion.V5)

    def resample(self, chunks: np.ndarray, sr_in: int) -> np.ndarray:
        return self.any_engine().resample(chunks, sr_in)

    def close(self) -> None:
        with self._lock:
            for eng in self._engines.values():
                eng.close()
            self._engines.clear()


_default: Optional[EnginePool] = None
_default_lock = threading.Lock()


def default_pool() -> EnginePool:
    global _default
    with _default_lock:
        if _default is None:
            _default = EnginePool()
        return _default


def resolve_model_path(config: VADConfig) -> str:
    """VADProcessor._get_model_directory + get_model_filename (silero_model.py:706-721,
    config.py:242-249).  Without ``model_path`` the reference looks in its package ``models/``
    directory; this package ships the same tensors as ``weights/silero_v{4,5}_16k.svw``."""
    if config.model_path:
        return str(config.model_path / config.get_model_filename())
    return weights_io.packaged_blob_path(_VERSION_INT[config.model_version])


# --------------------------------------------------------------------------------------
class StreamBatch:
    """Many streams, one launch per frame period.

    ``add()`` opens a slot with its own thresholds; ``step(frames)`` advances every listed
    stream by one frame and returns ``(probs, events, seg_frames)`` — the state machine runs on
    the device.  Segment audio / callbacks are layered on top by
    :class:`cutter_vad_amd.core.silero_model.SegmentAssembler` if the caller wants WAV payloads.
    """

    def __init__(self, config: Optional[VADConfig] = None, device_id: Optional[int] = None,
                 max_streams: Optional[int] = None, pool: Optional[EnginePool] = None) -> None:
        self.config = config or VADConfig()
        self._pool = pool or default_pool()
        self.engine = self._pool.engine_for(resolve_model_path(self.config), self.config.model_version, device_id,
                                            max_streams, int(self.config.sample_rate))
        self.slots: List[int] = []

    def add(self, n: int = 1, config: Optional[VADConfig] = None) -> np.ndarray:
        cfg = config or self.config
        new = self.engine.open_streams(n)
        self.engine.set_thresholds_many(new, (cfg.vad_start_probability, cfg.vad_end_probability, cfg.voice_start_ratio,
                                              cfg.voice_end_ratio, cfg.voice_start_frame_count, cfg.voice_end_frame_count))
        self.slots.extend(int(s) for s in new)
        return new

    def remove(self, slots: Sequence[int]) -> None:
        for s in slots:
            self.engine.close_stream(int(s))
            self.slots.remove(int(s))

    def step(self, frames: np.ndarray, slots: Optional[Sequence[int]] = None):
        s = np.asarray(self.slots if slots is None else slots, dtype=np.int64)
        thr = 0.01 if self.config.enable_denoising else None
        return self.engine.step_events(s, frames, denoise=thr)

    def step_rates(self, segments, slots: Optional[Sequence[int]] = None):
        """One tick for streams whose audio arrives at 8 / 16 / 24 / 48 kHz (``VADConfig.auto_convert_sample_rate``; the
        reference's hook is a ``pass``, vad_wrapper.py:621-624): ``segments`` = [(chunks [n_k, 512 * sr_k / 16000], sr_k), ...]
        in the order of ``slots``.  Resample + step chained on the GPU (``vad_step_rates``) ->
        ``(probs, events, seg_frames)``."""
        s = np.asarray(self.slots if slots is None else slots, dtype=np.int64)
        thr = 0.01 if self.config.enable_denoising else None
        return self.engine.step_rates(segments, s, denoise=thr)

    def close(self) -> None:
        for s in list(self.slots):
            self.engine.close_stream(s)
        self.slots.clear()

"""MI355X-native batched Silero-VAD engine behind the ``real_time_vad`` surface.

    from cutter_vad_amd import VADWrapper, VADConfig, SampleRate, SileroModelVersion

mirrors ``from real_time_vad import ...`` (/root/reference/src/real_time_vad/__init__.py:35-42).
The per-frame hot path (denoise gate -> Silero V4/V5 -> probability -> hysteresis) runs as
hand-written HIP kernels for gfx950 in ``libvad_engine.so`` (C ABI: ``include/vad_engine.h``).
There is no CPU execution path: without the library or without an MI355X the engine raises.
"""

from .core.config import SampleRate, SileroModelVersion, VADConfig
from .core.exceptions import (AudioProcessingError, CallbackError, ConfigurationError, ModelInitializationError,
                              ModelNotFoundError, VADError)
from .core.vad_wrapper import VADWrapper
from .core.async_vad_wrapper import AsyncVADWrapper
from .engine import Engine
from .pool import EnginePool, StreamBatch, default_pool
from .utils.audio import AudioUtils
from .utils.wav_writer import WAVWriter

__version__ = "0.1.0"

__all__ = ["VADWrapper", "AsyncVADWrapper", "VADConfig", "SampleRate", "SileroModelVersion", "VADError", "ModelNotFoundError",
           "ConfigurationError", "AudioProcessingError", "ModelInitializationError", "CallbackError", "AudioUtils",
           "WAVWriter", "Engine", "EnginePool", "StreamBatch", "default_pool"]

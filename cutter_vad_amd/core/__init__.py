from .config import SampleRate, SileroModelVersion, VADConfig
from .exceptions import (AudioProcessingError, CallbackError, ConfigurationError, ModelInitializationError,
                         ModelNotFoundError, VADError)

__all__ = ["VADConfig", "SampleRate", "SileroModelVersion", "VADError", "ModelNotFoundError", "ConfigurationError",
           "AudioProcessingError", "ModelInitializationError", "CallbackError"]

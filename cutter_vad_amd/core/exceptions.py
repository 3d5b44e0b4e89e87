"""Exception tree of the VAD library.

Same class names, constructor signatures, ``error_code`` strings, attributes and message
formats as /root/reference/src/real_time_vad/core/exceptions.py:8-68, so callers and tests
written against the reference keep working (SURVEY §8 b, "Error convention").
"""

from __future__ import annotations

from typing import Optional

__all__ = ["VADError", "ModelNotFoundError", "ConfigurationError", "AudioProcessingError",
           "ModelInitializationError", "CallbackError"]


class VADError(Exception):
    """Root of every error raised by this package."""

    def __init__(self, message: str, error_code: Optional[str] = None) -> None:
        super().__init__(message)
        self.message, self.error_code = message, error_code

    def __str__(self) -> str:
        return f"[{self.error_code}] {self.message}" if self.error_code else self.message


class ModelNotFoundError(VADError):
    def __init__(self, model_path: str, message: Optional[str] = None) -> None:
        super().__init__(message or f"Silero model not found at path: {model_path}", "MODEL_NOT_FOUND")
        self.model_path = model_path


class ConfigurationError(VADError):
    def __init__(self, parameter: str, value: str, message: Optional[str] = None) -> None:
        super().__init__(message or f"Invalid configuration for parameter '{parameter}': {value}",
                         "CONFIGURATION_ERROR")
        self.parameter, self.value = parameter, value


class AudioProcessingError(VADError):
    def __init__(self, message: str, audio_data_info: Optional[str] = None) -> None:
        super().__init__(message, "AUDIO_PROCESSING_ERROR")
        self.audio_data_info = audio_data_info


class ModelInitializationError(VADError):
    def __init__(self, model_version: str, message: Optional[str] = None) -> None:
        super().__init__(message or f"Failed to initialize Silero model version: {model_version}",
                         "MODEL_INITIALIZATION_ERROR")
        self.model_version = model_version


class CallbackError(VADError):
    def __init__(self, callback_name: str, original_error: Exception) -> None:
        super().__init__(f"Error in callback '{callback_name}': {original_error}", "CALLBACK_ERROR")
        self.callback_name, self.original_error = callback_name, original_error

"""``VADConfig`` and the two enums, field-for-field compatible with the reference
(/root/reference/src/real_time_vad/core/config.py:15-269): same names, defaults, bounds,
``validate_assignment`` / ``extra="forbid"`` behaviour and the dict / YAML / env loaders.

Engine-only knobs (device ordinal, pool capacity) deliberately live OUTSIDE this model, in
``cutter_vad_amd.pool`` — ``extra="forbid"`` must keep rejecting unknown keys exactly as the
reference does.
"""

from __future__ import annotations

import os
from enum import Enum, IntEnum
from pathlib import Path
from typing import Any, Dict, Optional, Union

import yaml
from pydantic import BaseModel, ConfigDict, Field, field_validator


class SampleRate(IntEnum):
    SAMPLERATE_8 = 8000
    SAMPLERATE_16 = 16000
    SAMPLERATE_24 = 24000
    SAMPLERATE_48 = 48000


class SileroModelVersion(Enum):
    V4 = "v4"
    V5 = "v5"


_MODEL_FILES = {SileroModelVersion.V4: "silero_vad.onnx", SileroModelVersion.V5: "silero_vad_v5.onnx"}

# VAD_<KEY> -> (field, caster); config.py:183-212 of the reference
_TRUE = ("true", "1", "yes", "on")
_ENV = {
    "SAMPLE_RATE": ("sample_rate", int),
    "MODEL_VERSION": ("model_version", str),
    "MODEL_PATH": ("model_path", str),
    "START_PROBABILITY": ("vad_start_probability", float),
    "END_PROBABILITY": ("vad_end_probability", float),
    "VOICE_START_RATIO": ("voice_start_ratio", float),
    "VOICE_END_RATIO": ("voice_end_ratio", float),
    "VOICE_START_FRAME_COUNT": ("voice_start_frame_count", int),
    "VOICE_END_FRAME_COUNT": ("voice_end_frame_count", int),
    "ENABLE_DENOISING": ("enable_denoising", lambda s: s.lower() in _TRUE),
    "AUTO_CONVERT_SAMPLE_RATE": ("auto_convert_sample_rate", lambda s: s.lower() in _TRUE),
    "BUFFER_SIZE": ("buffer_size", int),
}


class VADConfig(BaseModel):
    """All user-facing parameters of one VAD stream."""

    model_config = ConfigDict(use_enum_values=False, validate_assignment=True, extra="forbid")

    sample_rate: SampleRate = Field(default=SampleRate.SAMPLERATE_16, description="Audio sample rate for processing")
    model_version: SileroModelVersion = Field(default=SileroModelVersion.V5, description="Silero model version to use")
    model_path: Optional[Path] = Field(default=None, description="Custom directory holding the model files")

    vad_start_probability: float = Field(default=0.7, ge=0.0, le=1.0)
    vad_end_probability: float = Field(default=0.7, ge=0.0, le=1.0)
    voice_start_ratio: float = Field(default=0.8, ge=0.0, le=1.0)
    voice_end_ratio: float = Field(default=0.95, ge=0.0, le=1.0)
    voice_start_frame_count: int = Field(default=10, ge=1)   # 10 x 32 ms = 320 ms
    voice_end_frame_count: int = Field(default=50, ge=1)     # 50 x 32 ms = 1.6 s

    enable_denoising: bool = Field(default=True)
    auto_convert_sample_rate: bool = Field(default=True)
    buffer_size: int = Field(default=512, ge=256, le=2048)

    output_wav_sample_rate: int = Field(default=16000)
    output_wav_bit_depth: int = Field(default=16)

    @field_validator("model_path")
    @classmethod
    def _model_dir_must_exist(cls, v: Optional[Path]) -> Optional[Path]:
        if v is not None:
            if not v.exists():
                raise ValueError(f"Model path does not exist: {v}")
            if not v.is_dir():
                raise ValueError(f"Model path must be a directory: {v}")
        return v

    # ------------------------------------------------------------------ loaders
    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "VADConfig":
        d = config_dict  # the reference coerces in place, callers may rely on it
        sr = d.get("sample_rate")
        if isinstance(sr, str):
            d["sample_rate"] = getattr(SampleRate, f"SAMPLERATE_{sr}")
        elif isinstance(sr, int) and "sample_rate" in d:
            d["sample_rate"] = SampleRate(sr)
        if isinstance(d.get("model_version"), str):
            d["model_version"] = SileroModelVersion(d["model_version"].lower())
        if d.get("model_path"):
            d["model_path"] = Path(d["model_path"])
        return cls(**d)

    @classmethod
    def from_yaml(cls, yaml_path: Union[str, Path]) -> "VADConfig":
        yaml_path = Path(yaml_path)
        if not yaml_path.exists():
            raise FileNotFoundError(f"Configuration file not found: {yaml_path}")
        with open(yaml_path, "r", encoding="utf-8") as f:
            return cls.from_dict(yaml.safe_load(f))

    @classmethod
    def from_env(cls, prefix: str = "VAD_") -> "VADConfig":
        found: Dict[str, Any] = {}
        for key, (field, cast) in _ENV.items():
            raw = os.getenv(prefix + key)
            if raw is not None:
                found[field] = cast(raw)
        return cls.from_dict(found) if found else cls()

    # ------------------------------------------------------------------ dumps
    def to_dict(self) -> Dict[str, Any]:
        return self.model_dump()

    def _to_serializable_dict(self) -> Dict[str, Any]:
        data = self.model_dump()
        if isinstance(data.get("sample_rate"), SampleRate):
            data["sample_rate"] = data["sample_rate"].value
        if isinstance(data.get("model_version"), SileroModelVersion):
            data["model_version"] = data["model_version"].value
        if data.get("model_path") is not None:
            data["model_path"] = str(data["model_path"])
        return data

    def to_yaml(self, yaml_path: Union[str, Path]) -> None:
        yaml_path = Path(yaml_path)
        yaml_path.parent.mkdir(parents=True, exist_ok=True)
        with open(yaml_path, "w", encoding="utf-8") as f:
            yaml.dump(self._to_serializable_dict(), f, default_flow_style=False)

    # ------------------------------------------------------------------ helpers
    def get_model_filename(self) -> str:
        return _MODEL_FILES.get(self.model_version, "silero_vad.onnx")

    def get_frame_duration_ms(self) -> float:
        return (self.buffer_size / self.sample_rate) * 1000

    def __str__(self) -> str:
        return (f"VADConfig(\tsample_rate={self.sample_rate}Hz, \tmodel={self.model_version.value}, "
                f"\tstart_prob={self.vad_start_probability}, \tend_prob={self.vad_end_probability})")

    __repr__ = __str__

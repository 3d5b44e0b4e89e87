"""asyncio façade over :class:`VADWrapper` — the surface of
/root/reference/src/real_time_vad/core/async_vad_wrapper.py:22-379.

Same contract as the reference: every ``*_async`` method runs the synchronous wrapper method on a
small thread pool (``max_workers`` = 2 by default, :56) and awaits it; the synchronous methods
delegate directly; voice callbacks are coroutines that the worker thread posts onto the caller's
event loop with ``run_coroutine_threadsafe`` (:81-106), i.e. fire-and-forget from the point of view
of frame processing, and a coroutine that raises is re-raised as ``CallbackError`` inside its own
task (:125-134).  The loop is captured on first use: the running loop if there is one, otherwise a
fresh loop that nobody runs (:114-123 — callbacks posted from purely synchronous use never execute,
as in the reference).

The engine underneath is the shared MI355X stream pool: the executor threads only wait on the GPU,
so two workers are plenty for one stream; many streams belong in ``StreamBatch`` / ``SharedPoolServer``.
"""

from __future__ import annotations

import asyncio
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Awaitable, Callable, Dict, List, Optional, Union

import numpy as np

from .config import SampleRate, SileroModelVersion, VADConfig
from .exceptions import CallbackError
from .vad_wrapper import VADWrapper

AsyncVoiceStartCallback = Callable[[], Awaitable[None]]
AsyncVoiceEndCallback = Callable[[bytes], Awaitable[None]]
AsyncVoiceContinueCallback = Callable[[bytes], Awaitable[None]]


class AsyncVADWrapper:
    def __init__(self, config: Optional[VADConfig] = None, max_workers: int = 2) -> None:
        self.config = config if config is not None else VADConfig()
        self.vad_wrapper = VADWrapper(self.config)
        self.executor = ThreadPoolExecutor(max_workers=max_workers)
        self._async_voice_start_callback: Optional[AsyncVoiceStartCallback] = None
        self._async_voice_end_callback: Optional[AsyncVoiceEndCallback] = None
        self._async_voice_continue_callback: Optional[AsyncVoiceContinueCallback] = None
        self._callback_loop: Optional[asyncio.AbstractEventLoop] = None
        self._closed = False
        self._setup_sync_callbacks()

    # ------------------------------------------------------------------ callback bridge
    def _setup_sync_callbacks(self) -> None:
        """The wrapper calls these on the worker thread; each posts the user's coroutine to the loop."""

        def post(name: str, attr: str, *payload) -> None:
            cb = getattr(self, attr)
            if cb is not None:
                asyncio.run_coroutine_threadsafe(self._handle_async_callback(lambda: cb(*payload), name),
                                                 self._get_event_loop())

        self.vad_wrapper.set_callbacks(
            voice_start_callback=lambda: post("voice_start", "_async_voice_start_callback"),
            voice_end_callback=lambda wav: post("voice_end", "_async_voice_end_callback", wav),
            voice_continue_callback=lambda pcm: post("voice_continue", "_async_voice_continue_callback", pcm))

    def _get_event_loop(self) -> asyncio.AbstractEventLoop:
        if self._callback_loop is None or self._callback_loop.is_closed():
            try:
                self._callback_loop = asyncio.get_running_loop()
            except RuntimeError:
                self._callback_loop = asyncio.new_event_loop()
        return self._callback_loop

    async def _handle_async_callback(self, callback: Callable[[], Awaitable[None]], callback_name: str) -> None:
        try:
            await callback()
        except Exception as e:
            raise CallbackError(callback_name, e)

    def set_async_callbacks(self, voice_start_callback: Optional[AsyncVoiceStartCallback] = None,
                            voice_end_callback: Optional[AsyncVoiceEndCallback] = None,
                            voice_continue_callback: Optional[AsyncVoiceContinueCallback] = None) -> None:
        self._async_voice_start_callback = voice_start_callback
        self._async_voice_end_callback = voice_end_callback
        self._async_voice_continue_callback = voice_continue_callback

    # ------------------------------------------------------------------ awaitable operations
    async def _offload(self, fn: Callable[..., Any], *args) -> Any:
        loop = asyncio.get_running_loop()
        if self._callback_loop is None or self._callback_loop.is_closed():
            self._callback_loop = loop          # callbacks come back to the loop that submitted the audio
        return await loop.run_in_executor(self.executor, fn, *args)

    async def set_sample_rate_async(self, sample_rate: SampleRate) -> None:
        await self._offload(self.vad_wrapper.set_sample_rate, sample_rate)

    async def set_silero_model_async(self, model_version: SileroModelVersion) -> None:
        await self._offload(self.vad_wrapper.set_silero_model, model_version)

    async def set_thresholds_async(self, vad_start_probability: float = 0.7, vad_end_probability: float = 0.7,
                                   voice_start_ratio: float = 0.8, voice_end_ratio: float = 0.95,
                                   voice_start_frame_count: int = 10, voice_end_frame_count: int = 57) -> None:
        await self._offload(self.vad_wrapper.set_thresholds, vad_start_probability, vad_end_probability,
                            voice_start_ratio, voice_end_ratio, voice_start_frame_count, voice_end_frame_count)

    async def process_audio_data_async(self, audio_data: Union[np.ndarray, List[float]]) -> None:
        await self._offload(self.vad_wrapper.process_audio_data, audio_data)

    async def process_audio_data_with_buffer_async(self, audio_buffer: np.ndarray, count: int) -> None:
        await self._offload(self.vad_wrapper.process_audio_data_with_buffer, audio_buffer, count)

    async def reset_async(self) -> None:
        await self._offload(self.vad_wrapper.reset)

    async def get_statistics_async(self) -> Dict[str, Any]:
        return await self._offload(self.vad_wrapper.get_statistics)

    async def is_voice_active_async(self) -> bool:
        return await self._offload(self.vad_wrapper.is_voice_active)

    async def update_config_async(self, config: VADConfig) -> None:
        await self._offload(self.vad_wrapper.update_config, config)

    # ------------------------------------------------------------------ synchronous pass-throughs
    def set_sample_rate(self, sample_rate: SampleRate) -> None:
        self.vad_wrapper.set_sample_rate(sample_rate)

    def set_silero_model(self, model_version: SileroModelVersion) -> None:
        self.vad_wrapper.set_silero_model(model_version)

    def set_thresholds(self, vad_start_probability: float = 0.7, vad_end_probability: float = 0.7,
                       voice_start_ratio: float = 0.8, voice_end_ratio: float = 0.95,
                       voice_start_frame_count: int = 10, voice_end_frame_count: int = 57) -> None:
        self.vad_wrapper.set_thresholds(vad_start_probability, vad_end_probability, voice_start_ratio, voice_end_ratio,
                                        voice_start_frame_count, voice_end_frame_count)

    def process_audio_data(self, audio_data: Union[np.ndarray, List[float]]) -> None:
        self.vad_wrapper.process_audio_data(audio_data)

    def reset(self) -> None:
        self.vad_wrapper.reset()

    def get_statistics(self) -> Dict[str, Any]:
        return self.vad_wrapper.get_statistics()

    def is_voice_active(self) -> bool:
        return self.vad_wrapper.is_voice_active()

    def get_config(self) -> VADConfig:
        return self.vad_wrapper.get_config()

    def update_config(self, config: VADConfig) -> None:
        self.vad_wrapper.update_config(config)

    # ------------------------------------------------------------------ lifetime
    def cleanup(self) -> None:
        if self._closed:
            return
        self._closed = True
        self.vad_wrapper.cleanup()
        self.executor.shutdown(wait=True)

    async def acleanup(self) -> None:
        await asyncio.get_running_loop().run_in_executor(None, self.cleanup)

    def __enter__(self) -> "AsyncVADWrapper":
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        self.cleanup()

    async def __aenter__(self) -> "AsyncVADWrapper":
        return self

    async def __aexit__(self, exc_type, exc_val, exc_tb) -> None:
        await self.acleanup()

    def __del__(self) -> None:
        try:
            self.cleanup()
        except Exception:
            pass

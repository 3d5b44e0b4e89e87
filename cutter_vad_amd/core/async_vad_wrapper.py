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
import functools
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Awaitable, Callable, Optional

from .config import VADConfig
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
    async def _offload(self, fn: Callable[..., Any], *args, **kwargs) -> Any:
        loop = asyncio.get_running_loop()
        if self._callback_loop is None or self._callback_loop.is_closed():
            self._callback_loop = loop          # callbacks come back to the loop that submitted the audio
        return await loop.run_in_executor(self.executor, functools.partial(fn, *args, **kwargs))

    # The awaitable operations and the synchronous pass-throughs are generated below from two name tables: every one of them
    # is "the wrapper's method of the same name" - on the executor for ``<name>_async``, inline otherwise - with the wrapper
    # method's own signature, defaults and docstring (``functools.wraps``), so the two surfaces cannot drift apart.

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


# wrapper method -> which forms the facade has (the reference's surface: async_vad_wrapper.py:137-340)
_BOTH = ("set_sample_rate", "set_silero_model", "set_thresholds", "process_audio_data", "reset", "get_statistics", "is_voice_active",
         "update_config")
_SYNC_ONLY = ("get_config",)
_ASYNC_ONLY = ("process_audio_data_with_buffer",)


def _inline(name: str):
    @functools.wraps(getattr(VADWrapper, name))
    def call(self, *args, **kwargs):
        return getattr(self.vad_wrapper, name)(*args, **kwargs)
    return call


def _awaitable(name: str):
    @functools.wraps(getattr(VADWrapper, name))
    async def call(self, *args, **kwargs):
        return await self._offload(getattr(self.vad_wrapper, name), *args, **kwargs)
    call.__name__ = call.__qualname__ = f"{name}_async"
    return call


for _name in _BOTH + _SYNC_ONLY:
    setattr(AsyncVADWrapper, _name, _inline(_name))
for _name in _BOTH + _ASYNC_ONLY:
    setattr(AsyncVADWrapper, f"{_name}_async", _awaitable(_name))
del _name

"""ASGI application with the reference server's wire protocol on the shared stream pool.

Protocol (websocket_service/server/vad_websocket_server.py): ``GET /`` / ``/health`` / ``/clients`` (:751-788); websocket ``/vad`` with
query parameters mode / sample_rate / channels / sample_width / frame_duration_ms / start_probability /
end_probability / start_frame_count / end_frame_count / start_ratio / end_ratio / timeout (:532-548, :551-611);
binary messages = one PCM frame of exactly ``sample_rate * frame_duration_ms/1000 * channels * sample_width``
bytes, int16 little-endian scaled by 1/32767 or float32 (:326-347); text messages ``{"type": "CONFIG", ...}`` and
``{"type": "HEARTBEAT"}`` (:702-722); events as JSON objects ``event`` / ``timestamp_ms`` / ``segment_index``
(+ ``segment_start_ms`` / ``segment_end_ms`` / ``duration_ms`` on VOICE_END, ``message`` on INFO / ERROR /
TIMEOUT) (:83-124).

What differs from the reference is only where the model runs: the socket's receive loop just queues the frame
on its :class:`PooledSession`; one ticker task steps ALL sessions per tick with a single launch and posts the
events back to each socket.  Opus / AAC need PyAV, which this image does not have: those modes are refused with
the reference's own message (:645-655).
"""

from __future__ import annotations

import asyncio
import logging
import contextlib
import json
import threading
import time
import uuid
import weakref
from typing import Any, Dict, Optional, Sequence
from urllib.parse import parse_qs

import numpy as np
from fastapi import FastAPI, WebSocket, WebSocketDisconnect

from ..core.config import SampleRate, SileroModelVersion, VADConfig
from .shared_pool import PooledSession, SharedStreamPool
from .sharded_pool import ShardedStreamPool

_INT_KEYS = ("sample_rate", "channels", "sample_width", "frame_duration_ms", "start_frame_count", "end_frame_count")
_FLOAT_KEYS = ("start_probability", "end_probability", "timeout")
_AUDIO_KEYS = ("mode", "sample_rate", "channels", "sample_width", "frame_duration_ms")
_VAD_KEYS = ("start_probability", "end_probability", "start_frame_count", "end_frame_count", "start_ratio", "end_ratio")
_CONFIG_FIELDS = _AUDIO_KEYS + _VAD_KEYS + ("timeout",)


def now_ms() -> int:
    return int(time.time() * 1000)


def parse_query_params(query_string: str) -> Dict[str, Any]:
    """First value of each key; ints / floats for the keys the reference converts (:532-548) — start_ratio and
    end_ratio stay strings there, and so they do here (pydantic coerces them later)."""
    out: Dict[str, Any] = {}
    for key, values in parse_qs(query_string or "").items():
        if not values:
            continue
        v = values[0]
        out[key] = int(v) if key in _INT_KEYS else float(v) if key in _FLOAT_KEYS else v
    return out


def default_client_config() -> Dict[str, Any]:
    return {"audio": {"mode": "pcm", "sample_rate": 16000, "channels": 1, "sample_width": 2, "frame_duration_ms": 30},
            "vad": {"start_probability": 0.4, "end_probability": 0.3, "start_frame_count": 6, "end_frame_count": 12,
                    "start_ratio": 0.8, "end_ratio": 0.95},
            "timeout": 0.0}


def create_client_config(query_params: Dict[str, Any], config_message: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Defaults <- query parameters <- CONFIG message (:551-611)."""
    cfg = default_client_config()
    for layer in (query_params, {k: v for k, v in (config_message or {}).items() if v is not None}):
        for k, v in layer.items():
            if k in _AUDIO_KEYS:
                cfg["audio"][k] = v
            elif k in _VAD_KEYS:
                cfg["vad"][k] = v
            elif k == "timeout":
                cfg["timeout"] = v
    a, v = cfg["audio"], cfg["vad"]
    a["mode"] = str(a["mode"])
    for k in ("sample_rate", "channels", "sample_width", "frame_duration_ms"):
        a[k] = int(a[k])
    for k in ("start_probability", "end_probability", "start_ratio", "end_ratio"):
        v[k] = float(v[k])
    for k in ("start_frame_count", "end_frame_count"):
        v[k] = int(v[k])
    cfg["timeout"] = float(cfg["timeout"])
    return cfg


def frame_bytes(cfg: Dict[str, Any]) -> float:
    a = cfg["audio"]
    return a["sample_rate"] * (a["frame_duration_ms"] / 1000) * a["channels"] * a["sample_width"]


def vad_config_of(cfg: Dict[str, Any]) -> VADConfig:
    """ClientState._initialize_vad (:245-266): V5, denoising on, buffer_size = the client's frame length."""
    a, v = cfg["audio"], cfg["vad"]
    return VADConfig(sample_rate=SampleRate(a["sample_rate"]), model_version=SileroModelVersion.V5,
                     vad_start_probability=v["start_probability"], vad_end_probability=v["end_probability"],
                     voice_start_ratio=v["start_ratio"], voice_end_ratio=v["end_ratio"],
                     voice_start_frame_count=v["start_frame_count"], voice_end_frame_count=v["end_frame_count"],
                     enable_denoising=True, auto_convert_sample_rate=True,
                     buffer_size=int(a["sample_rate"] * (a["frame_duration_ms"] / 1000)))


class LoopRelay:
    """Pool callbacks run on a ticker's thread; the sockets live on an event loop.  ``call_soon_threadsafe`` per event wakes the
    loop through its self-pipe every time - a tick of a busy pool produces thousands of events (one VOICE_CONTINUE per talking
    client and frame).  The relay queues them and wakes the loop ONCE per batch: the first event of an empty queue schedules a
    drain on the loop, which takes whatever has arrived by then, in order."""

    def __init__(self, loop: asyncio.AbstractEventLoop) -> None:
        self._loop = weakref.ref(loop)           # weak: the relay is the VALUE of a WeakKeyDictionary keyed by the loop - a strong
        self._items: list = []                   # reference from here would keep every loop ever used (and its relay) alive
        self._lock = threading.Lock()

    @property
    def loop(self) -> Optional[asyncio.AbstractEventLoop]:
        return self._loop()

    def post(self, fn, *args) -> None:
        with self._lock:
            self._items.append((fn, args))
            first = len(self._items) == 1
        if first:
            loop = self._loop()
            if loop is None:
                raise RuntimeError("event loop is gone")
            loop.call_soon_threadsafe(self._drain)

    def _drain(self) -> None:
        with self._lock:
            items, self._items = self._items, []
        for fn, args in items:
            try:
                fn(*args)
            except Exception:
                logging.getLogger(__name__).exception("event delivery failed")


_RELAYS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()      # one relay per event loop, gone with it


def relay_of(loop: asyncio.AbstractEventLoop) -> LoopRelay:
    r = _RELAYS.get(loop)
    if r is None:
        r = _RELAYS[loop] = LoopRelay(loop)
    return r


class ClientSession:
    """One websocket client: wire decoding, event encoding, timeout; the audio goes to the pool."""

    def __init__(self, client_id: str, websocket: WebSocket, cfg: Dict[str, Any], pool: SharedStreamPool,
                 loop: asyncio.AbstractEventLoop) -> None:
        self.client_id = client_id
        self.websocket = websocket
        self.cfg = cfg
        self.pool = pool
        self.loop = loop
        self.start_time = time.time()
        self.segment_index = 0
        self.sent = 0                 # frames handed to the pool since the stream (re)started; backlog = sent - frames stepped
        self.voice_start_time: Optional[float] = None
        self.last_voice_time: Optional[float] = None
        self.timeout_task: Optional[asyncio.Task] = None
        self.expected_frame_bytes = int(frame_bytes(cfg)) if cfg["audio"]["mode"] == "pcm" else 0
        self.outbox: "asyncio.Queue[Optional[str]]" = asyncio.Queue()
        self.session: Optional[PooledSession] = None
        self.session_error: Optional[str] = None
        self._open(cfg)

    def _open(self, cfg: Dict[str, Any]) -> None:
        """The reference builds the wrapper for any SampleRate and only fails when a frame reaches the model
        (the V5 graph's 8 kHz branch cannot take 512-sample frames): same here, frame by frame."""
        try:
            vc = vad_config_of(cfg)
            if self.session is None:
                self.session = self.pool.open_session(vc)
                self._bind()
            else:
                self.pool.reconfigure(self.session, vc)
            self.sent = 0                                  # a (re)configured stream starts from a clean state, queue included
            self.session_error = None
        except Exception as e:
            self.session_error = f"Audio processing failed: Frame processing failed: {e}"

    def _bind(self) -> None:
        # pool callbacks run on the ticker's thread: hand the event to the socket's loop, in order, one wake-up per batch
        post = relay_of(self.loop).post
        self.session.set_callbacks(lambda: post(self._on_voice_start), lambda wav: post(self._on_voice_end),
                                   lambda pcm: post(self._on_voice_continue),
                                   lambda e: post(self.send_error, f"Audio processing error: {e}"),
                                   continue_payload=False)        # VOICE_CONTINUE carries no audio (:420-430)

    # -- events (:83-124, :428-498)
    def _emit(self, event: str, **fields) -> None:
        msg = {"event": event, "timestamp_ms": fields.pop("timestamp_ms", now_ms()), "segment_index": None}
        msg.update(fields)
        self.outbox.put_nowait(json.dumps(msg))

    def send_info(self, message: str) -> None:
        self._emit("INFO", message=message)

    def send_error(self, message: str) -> None:
        self._emit("ERROR", message=message)

    def _on_voice_start(self) -> None:
        self.voice_start_time = time.time()
        self._emit("VOICE_START", segment_index=self.segment_index)

    def _on_voice_continue(self) -> None:
        # one per frame and talking client: the text json.dumps would produce, without the dict and the encoder
        self.outbox.put_nowait('{"event": "VOICE_CONTINUE", "timestamp_ms": %d, "segment_index": %d}' % (now_ms(), self.segment_index))

    def _on_voice_end(self) -> None:
        t = time.time()
        end_ms = int(t * 1000)
        start_ms = int(self.voice_start_time * 1000) if self.voice_start_time else end_ms
        self._emit("VOICE_END", timestamp_ms=end_ms, segment_index=self.segment_index, segment_start_ms=start_ms,
                   segment_end_ms=end_ms, duration_ms=end_ms - start_ms)
        self.segment_index += 1
        if self.timeout_task:
            self.timeout_task.cancel()
            self.timeout_task = None
        if self.cfg["timeout"] > 0:
            self.timeout_task = asyncio.ensure_future(self._timeout_monitor())

    async def _timeout_monitor(self) -> None:
        try:
            while True:
                await asyncio.sleep(1.0)
                if self.last_voice_time and self.cfg["timeout"] > 0 and time.time() - self.last_voice_time >= self.cfg["timeout"]:
                    self._emit("TIMEOUT", message="no voice detected in configured timeout")
                    break
        except asyncio.CancelledError:
            pass

    # -- audio (:326-380)
    def process_audio_frame(self, data: bytes) -> None:
        a = self.cfg["audio"]
        if a["mode"] != "pcm":
            self.send_error(f"No decoder available for {a['mode']}")
            return
        if len(data) != self.expected_frame_bytes:
            self.send_error(f"Invalid frame size: expected {self.expected_frame_bytes}, got {len(data)}")
            return
        x = None
        if a["sample_width"] == 2:
            pass          # int16 little-endian: the bytes travel to the GPU as they are, scaled by 1/32767 in the kernel
        elif a["sample_width"] == 4:
            x = np.frombuffer(data, dtype=np.float32)
        else:
            self.send_error(f"Unsupported sample width: {a['sample_width']}")
            return
        try:
            # multi-channel frames go to the model interleaved, as the reference hands them over (:369)
            if self.session_error is not None or self.session is None:
                raise RuntimeError(self.session_error or "VAD wrapper not initialized")
            if x is None:
                self.session.submit_pcm16(data)
            else:
                self.session.submit(x)
            self.sent += 1
            self.last_voice_time = time.time()
            if self.cfg["timeout"] > 0 and not self.timeout_task:
                self.timeout_task = asyncio.ensure_future(self._timeout_monitor())
        except Exception as e:
            self.send_error(f"Audio processing error: {e}")

    def backlog(self) -> int:
        """frames this client has sent that the pool has neither stepped nor dropped (a failed tick, a refused push: `lost`)"""
        s = self.session
        return self.sent - s.frames_done - s.lost if s is not None and not s.closed else 0

    async def wait_for_pool(self, interval: float) -> None:
        """Back-pressure: do not read the socket until the pool is within BACKLOG_LOW frames of this client.  The wait ends with
        the session (closed, failed) and cannot outlive the pool's progress: STALL_INTERVALS intervals in which not one of the
        client's frames was stepped or dropped mean the count is off (frames the engine lost without saying whose), and the
        count is resynchronised instead of leaving the socket unread for good."""
        stalled, last = 0, -1
        while self.backlog() > BACKLOG_LOW:
            s = self.session
            if s is None or s.closed or self.session_error is not None:
                return
            moved = s.frames_done + s.lost
            stalled = stalled + 1 if moved == last else 0
            last = moved
            if stalled >= STALL_INTERVALS:
                self.sent = moved
                return
            await asyncio.sleep(interval)

    def update_config(self, cfg: Dict[str, Any]) -> None:
        old, self.cfg = self.cfg, cfg
        if old["audio"] != cfg["audio"]:
            self.expected_frame_bytes = int(frame_bytes(cfg)) if cfg["audio"]["mode"] == "pcm" else 0
        if old["vad"] != cfg["vad"] or old["audio"]["sample_rate"] != cfg["audio"]["sample_rate"] \
                or old["audio"]["frame_duration_ms"] != cfg["audio"]["frame_duration_ms"]:
            self._open(cfg)

    def cleanup(self) -> None:
        if self.timeout_task:
            self.timeout_task.cancel()
        if self.session is not None:
            self.session.close()
        self.outbox.put_nowait(None)


BACKLOG_HIGH, BACKLOG_LOW = 96, 32          # frames a client may be ahead of the pool before / after its socket is paused
STALL_INTERVALS = 100                       # tick intervals without progress after which a paused socket is read again


def create_app(pool: Optional[SharedStreamPool] = None, tick_interval: float = 0.010, convert_rates: bool = False,
               devices: Optional[Sequence[int]] = None) -> FastAPI:
    """``convert_rates``: clients that announce 8 / 24 / 48 kHz with 32 ms frames are resampled on the GPU inside the pool's
    tick (SharedStreamPool); off, such a client gets the reference's per-frame error.
    ``devices``: HIP device ordinals - one stream pool (engine + ticker thread) per GPU behind this one app
    (``ShardedStreamPool``: a new client lands on the least-loaded GPU); a ready-made sharded pool may be passed as ``pool``."""
    state: Dict[str, Any] = {"pool": pool, "clients": {}, "ticker": None}

    @contextlib.asynccontextmanager
    async def lifespan(_app: FastAPI):
        yield
        if state["ticker"] is not None:
            state["ticker"].cancel()
        if state["pool"] is not None and hasattr(state["pool"], "shards"):
            state["pool"].stop()

    app = FastAPI(title="VAD WebSocket Server", description="Real-time Voice Activity Detection WebSocket Server "
                  "(shared MI355X stream pool)", version="1.0.0", lifespan=lifespan)
    app.state.vad = state

    def get_pool() -> SharedStreamPool:
        if state["pool"] is None:
            if devices is not None:
                state["pool"] = ShardedStreamPool(devices=devices, tick_interval=tick_interval, convert_rates=convert_rates)
            else:
                state["pool"] = SharedStreamPool(tick_interval=tick_interval, convert_rates=convert_rates)
        return state["pool"]

    async def ticker() -> None:
        loop = asyncio.get_running_loop()
        p = get_pool()
        while True:
            t0 = time.perf_counter()
            if p.session_count:
                try:
                    await loop.run_in_executor(None, p.tick)    # the launch + fan-out never block the event loop
                except asyncio.CancelledError:
                    raise
                except Exception:                               # one bad tick must not stop every client's VAD events
                    logging.getLogger(__name__).exception("tick failed")
            if getattr(p, "backlog", 0):                        # frames already staged for the next tick: catch up first
                await asyncio.sleep(0)
                continue
            await asyncio.sleep(max(0.0, tick_interval - (time.perf_counter() - t0)))

    def ensure_ticker() -> None:
        p = get_pool()
        if hasattr(p, "shards"):                 # one ticker thread conducts every GPU's tick (ShardedStreamPool.start)
            p.tick_interval = tick_interval
            p.start()
        elif state["ticker"] is None or state["ticker"].done():
            state["ticker"] = asyncio.ensure_future(ticker())

    @app.get("/")
    async def root():
        return {"message": "VAD WebSocket Server", "status": "running", "connected_clients": len(state["clients"]),
                "timestamp": now_ms()}

    @app.get("/health")
    async def health():
        return {"status": "healthy", "connected_clients": len(state["clients"]), "timestamp": now_ms()}

    @app.get("/clients")
    async def list_clients():
        info = {cid: {"connected_at": c.start_time, "config": c.cfg, "segment_index": c.segment_index}
                for cid, c in state["clients"].items()}
        return {"connected_clients": len(info), "clients": info, "timestamp": now_ms()}

    @app.get("/stats")
    async def stats():
        return get_pool().stats()

    async def refuse(websocket: WebSocket, message: str) -> None:
        await websocket.send_text(json.dumps({"event": "ERROR", "message": message, "timestamp_ms": now_ms()}))
        await websocket.close()

    @app.websocket("/vad")
    async def websocket_endpoint(websocket: WebSocket):
        client_id = str(uuid.uuid4())
        client: Optional[ClientSession] = None
        sender: Optional[asyncio.Task] = None
        try:
            await websocket.accept()
            query = parse_query_params(str(websocket.query_params))
            try:
                cfg = create_client_config(query)
            except (TypeError, ValueError) as e:
                await refuse(websocket, f"Invalid configuration: {e}")
                return
            mode = cfg["audio"]["mode"]
            if mode not in ("pcm", "opus", "aac"):
                await refuse(websocket, f"Unsupported audio mode: {mode}")
                return
            if mode in ("opus", "aac"):
                await refuse(websocket, f"PyAV is required for {mode} decoding but not installed")
                return
            fb = frame_bytes(cfg)
            if fb != int(fb):
                await refuse(websocket, f"Frame duration {cfg['audio']['frame_duration_ms']}ms produces non-integer bytes ({fb})")
                return
            ensure_ticker()
            client = ClientSession(client_id, websocket, cfg, get_pool(), asyncio.get_running_loop())
            state["clients"][client_id] = client

            async def pump() -> None:
                while True:
                    msg = await client.outbox.get()
                    if msg is None:
                        return
                    try:
                        await websocket.send_text(msg)
                    except Exception:
                        return

            sender = asyncio.ensure_future(pump())
            client.send_info("VAD WebSocket server ready")
            while True:
                try:
                    message = await websocket.receive()
                    if message.get("type") == "websocket.disconnect":
                        break
                    if message.get("bytes") is not None:
                        client.process_audio_frame(message["bytes"])
                        # Back-pressure.  The reference runs the model inside this loop (vad_websocket_server.py:369), so a client
                        # that sends faster than real time - a file, not a microphone - is simply read as fast as its frames are
                        # processed.  Here frames are queued and stepped one per tick: when a client is more than BACKLOG_HIGH
                        # frames ahead, its socket is not read until the pool has caught up (TCP then slows the sender), instead
                        # of running into the engine's 256-frames-waiting refusal.
                        if (client.sent & 15) == 0 and client.backlog() > BACKLOG_HIGH:
                            await client.wait_for_pool(tick_interval)
                    elif message.get("text") is not None:
                        try:
                            data = json.loads(message["text"])
                            kind = data.get("type") if isinstance(data, dict) else None
                            if kind == "CONFIG":
                                client.update_config(create_client_config(query, {k: data.get(k) for k in _CONFIG_FIELDS}))
                                client.send_info("Configuration updated")
                            elif kind == "HEARTBEAT":
                                client.send_info("Heartbeat received")
                            else:
                                client.send_error(f"Unknown message type: {kind}")
                        except json.JSONDecodeError as e:
                            client.send_error(f"Invalid JSON: {e}")
                        except (TypeError, ValueError) as e:
                            client.send_error(f"Invalid message format: {e}")
                except WebSocketDisconnect:
                    break
                except Exception as e:
                    client.send_error(f"Server error: {e}")
                    break
        finally:
            if client is not None:
                client.cleanup()
                state["clients"].pop(client_id, None)
            if sender is not None:
                try:
                    await asyncio.wait_for(sender, 1.0)
                except Exception:
                    sender.cancel()

    return app

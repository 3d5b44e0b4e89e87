"""Shared-pool serving layer on CPU (scripted engine): batching, per-session semantics, wire protocol."""

import json

import numpy as np
import pytest

from cutter_vad_amd import VADConfig
from cutter_vad_amd.core.exceptions import AudioProcessingError
from cutter_vad_amd.server import SharedStreamPool
from tests.fakes import FakeEngine, FakePool

LOUD = np.full(480, 0.5, np.float32)
QUIET = np.zeros(480, np.float32)


def make_pool():
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    calls = []
    orig = eng.step_events

    def counting(slots, frames, denoise=0.01, i16_scale=32767):
        calls.append(len(slots))
        return orig(slots, frames, denoise, i16_scale)

    eng.step_events = counting
    return SharedStreamPool(pool=FakePool(eng)), eng, calls


def test_one_launch_per_tick_and_reference_semantics():
    pool, eng, calls = make_pool()
    cfg = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3,
               buffer_size=480)
    sessions, logs = [], []
    for k in range(5):
        s = pool.open_session(VADConfig(**cfg))
        log = []
        s.set_callbacks(lambda log=log: log.append("S"), lambda wav, log=log: log.append(("E", len(wav))),
                        lambda pcm, log=log: log.append(("C", len(pcm))))
        sessions.append(s)
        logs.append(log)
    # session k talks from tick k on; 3 loud frames then silence.  Frames are queued ahead of the ticks.
    for k, s in enumerate(sessions):
        for f in [QUIET] * k + [LOUD] * 3 + [QUIET] * 4:
            s.submit(f)
    n = pool.drain()
    assert n == sum(7 + k for k in range(5))
    assert calls[:7] == [5] * 7 and calls[7:] == [4, 3, 2, 1]          # one launch per tick for everyone pending
    for log in logs:
        # START on the 2nd loud frame, CONTINUE for loud #3 and the 3 quiet frames, END on the 3rd quiet one:
        # on the END frame voice_end comes before voice_continue (core/vad_wrapper.py:505-519)
        assert log == ["S", ("C", 1920), ("C", 1920), ("C", 1920), ("E", 44 + 2 * 480 * 6), ("C", 1920)]
    assert eng.frames_seen[0].shape == (5, 512) and np.all(eng.frames_seen[0][:, 480:] == 0)   # right zero-pad to 512
    assert pool.stats()["frames_per_launch"] > 3
    # per-session order and independence: a second utterance only on session 2
    for f in [LOUD] * 2:
        sessions[2].submit(f)
    pool.drain()
    assert logs[2][-1] == "S" and sessions[2].is_voice_active() and not sessions[1].is_voice_active()
    # a failing callback reaches the session's error hook, the other sessions carry on
    errs = []
    sessions[2].set_callbacks(voice_continue_callback=lambda b: 1 / 0, error_callback=errs.append)
    sessions[2].submit(LOUD)
    sessions[3].submit(LOUD)
    assert pool.tick() == 2 and len(errs) == 1 and "voice_continue" in str(errs[0])
    # validation mirrors AudioUtils.validate_audio_data
    with pytest.raises(AudioProcessingError, match="NaN"):
        sessions[0].submit(np.array([np.nan] * 480, np.float32))
    with pytest.raises(AudioProcessingError, match="empty"):
        sessions[0].submit(np.zeros(0, np.float32))
    sessions[0].close()
    with pytest.raises(AudioProcessingError, match="closed"):
        sessions[0].submit(LOUD)
    assert pool.session_count == 4 and eng.closed == [sessions[0].slot]
    pool.reconfigure(sessions[2], VADConfig(**{**cfg, "voice_start_frame_count": 1}))
    assert not sessions[2].is_voice_active() and eng.thr[sessions[2].slot][4] == 1
    # the callbacks bound at open survive a reconfigure, voice_continue included (the app binds them once per client)
    after = []
    sessions[2].set_callbacks(lambda: after.append("S"), lambda wav: after.append("E"), lambda pcm: after.append("C"))
    pool.reconfigure(sessions[2], VADConfig(**{**cfg, "voice_start_frame_count": 1}))
    for f in [LOUD] * 3 + [QUIET] * 3:
        sessions[2].submit(f)
    pool.drain()
    assert after == ["S", "C", "C", "C", "C", "E", "C"]
    pool.close()
    assert pool.session_count == 0


def test_pcm16_wire_frames_travel_as_int16_and_formats_are_grouped():
    pool, eng, calls = make_pool()
    cfg = VADConfig(voice_start_frame_count=1, buffer_size=480)
    a, b, c = pool.open_session(cfg), pool.open_session(cfg), pool.open_session(cfg)
    loud16 = (np.full(480, 0.5) * 32767).astype("<i2").tobytes()
    got = []
    for s in (a, b, c):
        s.set_callbacks(voice_start_callback=lambda s=s: got.append(s.slot), voice_continue_callback=lambda pcm: got.append(len(pcm)))
    a.submit_pcm16(loud16)
    b.submit_pcm16(loud16)
    c.submit(LOUD)
    assert pool.tick() == 3 and sorted(calls) == [1, 2]             # one launch per wire format
    assert sorted(got) == [a.slot, b.slot, c.slot]
    i16_batch = [f for f in eng.frames_seen if f.shape[0] == 2][0]
    assert np.allclose(i16_batch[:, :480], 16383 / 32767.0) and np.all(i16_batch[:, 480:] == 0)
    a.submit_pcm16(loud16)
    a.submit_pcm16(loud16[:640])                                     # ragged lengths in one tick still work
    b.submit_pcm16(loud16[:640])
    assert pool.tick() == 2 and pool.tick() == 1
    assert got[-3:] == [1920, 1280, 1280] and a.frames_done == 3 and abs(a.last_probability - 0.9) < 1e-6
    with pytest.raises(AudioProcessingError):
        a.submit_pcm16(b"\0")
    pool.close()


def test_background_ticker_thread():
    import time
    pool, eng, calls = make_pool()
    pool.tick_interval = 0.002
    s = pool.open_session(VADConfig(voice_start_frame_count=1, buffer_size=480))
    got = []
    s.set_callbacks(voice_start_callback=lambda: got.append("S"))
    pool.start()
    s.submit(LOUD)
    t0 = time.time()
    while not got and time.time() - t0 < 5:
        time.sleep(0.005)
    pool.close()
    assert got == ["S"]


def test_wire_protocol_over_asgi():
    from fastapi.testclient import TestClient
    from cutter_vad_amd.server.app import create_app, create_client_config, parse_query_params
    q = parse_query_params("sample_rate=16000&start_probability=0.5&start_ratio=0.7&mode=pcm&x=1")
    assert q == {"sample_rate": 16000, "start_probability": 0.5, "start_ratio": "0.7", "mode": "pcm", "x": "1"}
    cfg = create_client_config(q, {"end_frame_count": 4, "timeout": None})
    assert cfg["vad"]["start_ratio"] == 0.7 and cfg["vad"]["end_frame_count"] == 4 and cfg["audio"]["frame_duration_ms"] == 30

    pool, eng, calls = make_pool()
    app = create_app(pool, tick_interval=0.002)
    loud = (np.full(480, 0.5) * 32767).astype("<i2").tobytes()
    quiet = np.zeros(480, "<i2").tobytes()

    def recv_until(ws, kind, limit=200):
        seen = []
        for _ in range(limit):
            m = json.loads(ws.receive_text())
            seen.append(m)
            if m["event"] == kind:
                return seen
        raise AssertionError(f"no {kind} in {seen}")

    with TestClient(app) as client:
        assert client.get("/").json()["status"] == "running"
        with client.websocket_connect("/vad?start_frame_count=2&end_frame_count=3") as ws:
            hello = json.loads(ws.receive_text())
            assert hello["event"] == "INFO" and hello["message"] == "VAD WebSocket server ready" and "timestamp_ms" in hello
            assert client.get("/").json()["connected_clients"] == 1
            assert client.get("/health").json()["status"] == "healthy"
            info = client.get("/clients").json()
            assert info["connected_clients"] == 1
            (only,) = info["clients"].values()
            assert only["segment_index"] == 0 and only["config"]["vad"]["start_frame_count"] == 2
            ws.send_text(json.dumps({"type": "HEARTBEAT"}))
            assert json.loads(ws.receive_text())["message"] == "Heartbeat received"
            ws.send_bytes(b"\0" * 100)
            m = json.loads(ws.receive_text())
            assert m["event"] == "ERROR" and m["message"] == "Invalid frame size: expected 960, got 100"
            for _ in range(3):
                ws.send_bytes(loud)
            seen = recv_until(ws, "VOICE_CONTINUE")
            assert [m["event"] for m in seen] == ["VOICE_START", "VOICE_CONTINUE"] and seen[0]["segment_index"] == 0
            for _ in range(3):
                ws.send_bytes(quiet)
            seen = recv_until(ws, "VOICE_END")
            end = seen[-1]
            assert end["segment_index"] == 0 and end["duration_ms"] == end["segment_end_ms"] - end["segment_start_ms"] >= 0
            assert all(m["event"] == "VOICE_CONTINUE" for m in seen[:-1])
            ws.send_text(json.dumps({"type": "CONFIG", "start_frame_count": 1, "frame_duration_ms": 20}))
            assert recv_until(ws, "INFO")[-1]["message"] == "Configuration updated"
            ws.send_bytes((np.full(320, 0.5) * 32767).astype("<i2").tobytes())          # 20 ms frames now
            seen = recv_until(ws, "VOICE_START")
            assert seen[-1]["segment_index"] == 1
            ws.send_bytes((np.full(320, 0.5) * 32767).astype("<i2").tobytes())
            assert recv_until(ws, "VOICE_CONTINUE")[-1]["event"] == "VOICE_CONTINUE"   # still delivered after a second CONFIG
            ws.send_text("{not json")
            assert recv_until(ws, "ERROR")[-1]["message"].startswith("Invalid JSON")
            ws.send_text(json.dumps({"type": "NOPE"}))
            assert recv_until(ws, "ERROR")[-1]["message"] == "Unknown message type: NOPE"
        with client.websocket_connect("/vad?mode=opus") as ws:
            m = json.loads(ws.receive_text())
            assert m["event"] == "ERROR" and "PyAV is required for opus" in m["message"]
        with client.websocket_connect("/vad?frame_duration_ms=1&sample_rate=16000&sample_width=2&channels=1") as ws:
            assert json.loads(ws.receive_text())["event"] == "INFO"       # 32 bytes: integral
        with client.websocket_connect("/vad?sample_rate=8000") as ws:
            assert json.loads(ws.receive_text())["event"] == "INFO"
            ws.send_bytes(np.zeros(240, "<i2").tobytes())
            m = json.loads(ws.receive_text())
            assert m["event"] == "ERROR" and m["message"].startswith("Audio processing error:")
        assert client.get("/stats").json()["launches"] >= 5
    assert pool.session_count == 0


def test_sessions_at_other_rates_are_resampled_inside_the_tick():
    """convert_rates: 8 / 24 / 48 kHz clients on a 16 kHz V5 pool (VADConfig.auto_convert_sample_rate, the hook the reference
    leaves empty: vad_wrapper.py:621-624).  Chunks of 32 ms at the client's rate; the model sees scipy's Fourier resample."""
    from oracle import oracle
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    pool = SharedStreamPool(pool=FakePool(eng), convert_rates=True)
    common = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=2)
    rng = np.random.default_rng(5)
    logs, sessions = {}, {}
    for sr, n_in in ((8000, 256), (24000, 768), (48000, 1536), (16000, 512)):
        s = pool.open_session(VADConfig(sample_rate=sr, buffer_size=n_in, **common))
        assert s.rate == (None if sr == 16000 else sr)
        logs[sr] = []
        s.set_callbacks(lambda sr=sr: logs[sr].append("S"), lambda wav, sr=sr: logs[sr].append(("E", len(wav))),
                        lambda pcm, sr=sr: logs[sr].append(("C", len(pcm))))
        sessions[sr] = (s, n_in)
    t = np.arange(1536 * 8) / 48000.0
    tone48 = (0.6 * np.sin(2 * np.pi * 300.0 * t)).astype(np.float32)
    sent = {}
    for sr, (s, n_in) in sessions.items():
        tone = tone48[::48000 // sr]                              # the same 300 Hz tone at every rate
        chunks = [np.zeros(n_in, np.float32)] + [tone[k * n_in:(k + 1) * n_in] for k in range(3)] + [np.zeros(n_in, np.float32)] * 3
        sent[sr] = chunks
        for k, c in enumerate(chunks):
            if sr == 24000 and k % 2:                             # int16 wire frames resample too
                s.submit_pcm16((c * 32767).astype("<i2").tobytes())
            else:
                s.submit(c)
    assert pool.drain() == 4 * 7
    # every client heard the same thing: START on the 2nd loud chunk, END on the 2nd quiet one; payloads at the client's rate
    for sr, (s, n_in) in sessions.items():
        assert logs[sr] == ["S", ("C", 4 * n_in), ("C", 4 * n_in), ("E", 44 + 2 * n_in * 5), ("C", 4 * n_in)], (sr, logs[sr])
    # what the model saw for the 48 kHz client's first loud chunk is the oracle's resample of it
    want = oracle.resample(sent[48000][1], 512)
    seen = [f for f in eng.frames_seen if f.shape[0] >= 1 and f.shape[1] == 512]
    assert any(np.array_equal(want, row) for f in seen for row in f)     # (the scripted engine records frames before the gate)
    assert pool.stats()["launches"] < pool.stats()["frames"]
    # a chunk of the wrong length is the session's error, not the pool's
    with pytest.raises(Exception, match="Failed to resample"):
        sessions[48000][0].submit(np.zeros(1440, np.float32))
    with pytest.raises(AudioProcessingError, match="buffer_size"):
        pool.open_session(VADConfig(sample_rate=48000, buffer_size=1440))
    pool.close()
    # without the flag such a session fails like the reference's V5 graph does
    plain = SharedStreamPool(pool=FakePool(FakeEngine(fn=lambda fr: 0.0)))
    with pytest.raises(AudioProcessingError, match="8 kHz graph branch"):
        plain.open_session(VADConfig(sample_rate=48000, buffer_size=1536))
    plain.close()


def test_websocket_client_at_48_khz_with_32_ms_frames():
    """create_app(convert_rates=True): a client announcing 48 kHz / 32 ms frames is resampled inside the pool's tick; with 30 ms
    frames (1440 samples, not one model frame's worth) it hears the reason on every frame, like any other per-frame failure."""
    from fastapi.testclient import TestClient
    from cutter_vad_amd.server.app import create_app
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    app = create_app(SharedStreamPool(pool=FakePool(eng), convert_rates=True), tick_interval=0.002)
    t = np.arange(1536) / 48000.0
    loud = (0.6 * np.sin(2 * np.pi * 300.0 * t) * 32767).astype("<i2").tobytes()

    def recv_until(ws, kind, limit=200):
        for _ in range(limit):
            m = json.loads(ws.receive_text())
            if m["event"] == kind:
                return m
        raise AssertionError(f"no {kind}")

    with TestClient(app) as client:
        with client.websocket_connect("/vad?sample_rate=48000&frame_duration_ms=32&start_frame_count=2&end_frame_count=2") as ws:
            assert json.loads(ws.receive_text())["event"] == "INFO"
            for _ in range(3):
                ws.send_bytes(loud)
            assert recv_until(ws, "VOICE_START")["segment_index"] == 0
            for _ in range(3):
                ws.send_bytes(b"\0" * len(loud))
            assert recv_until(ws, "VOICE_END")["segment_index"] == 0
        assert any(f.shape[1] == 512 and np.abs(f).max() > 0.5 for f in eng.frames_seen)
        with client.websocket_connect("/vad?sample_rate=48000&frame_duration_ms=30") as ws:
            assert json.loads(ws.receive_text())["event"] == "INFO"
            ws.send_bytes(b"\0" * 2880)
            m = recv_until(ws, "ERROR")
            assert "Failed to resample audio from 48000Hz" in m["message"] and "1536" in m["message"]


def test_reconfigure_moves_a_session_between_native_and_resampled_ingest():
    """A CONFIG message may change a client's sample rate (vad_websocket_server.py:300-318 rebuilds the wrapper): the pooled
    session switches between engine-rate frames and resampled chunks, starts from a clean state, and chunks of the old length
    are refused with the reason."""
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    pool = SharedStreamPool(pool=FakePool(eng), convert_rates=True)
    thr = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=1, voice_end_frame_count=2)
    s = pool.open_session(VADConfig(buffer_size=512, **thr))
    got = []
    s.set_callbacks(lambda: got.append("S"), lambda wav: got.append(("E", len(wav))))
    assert s.rate is None
    s.submit(np.full(512, 0.5, np.float32))
    assert pool.tick() == 1 and got == ["S"] and s.is_voice_active()
    pool.reconfigure(s, VADConfig(sample_rate=48000, buffer_size=1536, **thr))
    assert s.rate == 48000 and not s.is_voice_active()
    with pytest.raises(Exception, match="Failed to resample"):
        s.submit(np.full(512, 0.5, np.float32))
    t = np.arange(1536) / 48000.0
    s.submit((0.6 * np.sin(2 * np.pi * 300.0 * t)).astype(np.float32))
    s.submit(np.zeros(1536, np.float32))
    s.submit(np.zeros(1536, np.float32))
    assert pool.drain() == 3
    assert got == ["S", "S", ("E", 44 + 2 * 1536 * 3)]
    pool.reconfigure(s, VADConfig(buffer_size=480, **thr))
    assert s.rate is None
    s.submit(np.full(480, 0.5, np.float32))
    assert pool.tick() == 1 and got[-1] == "S"
    with pytest.raises(AudioProcessingError, match="buffer_size"):
        pool.reconfigure(s, VADConfig(sample_rate=24000, buffer_size=512, **thr))
    pool.close()


# ------------------------------------------------------------------ several pools behind one front (SURVEY §8 e)
def _sharded(n=2):
    from cutter_vad_amd.server import ShardedStreamPool
    engines = [FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05) for _ in range(n)]
    return ShardedStreamPool(shards=[SharedStreamPool(pool=FakePool(e)) for e in engines]), engines


def _script(k):
    """session k: quiet, an utterance, quiet, a second utterance"""
    return [QUIET] * (k % 3) + [LOUD] * 4 + [QUIET] * 4 + [LOUD] * 3 + [QUIET] * 4


def _run_sessions(pool, cfg, n, move=None):
    sessions, logs = [], []
    for k in range(n):
        s = pool.open_session(cfg)
        log = []
        s.set_callbacks(lambda log=log: log.append("S"), lambda wav, log=log: log.append(("E", bytes(wav))),
                        lambda pcm, log=log: log.append(("C", bytes(pcm))))
        sessions.append(s)
        logs.append(log)
    for t in range(max(len(_script(k)) for k in range(n))):
        for k, s in enumerate(sessions):
            sc = _script(k)
            if t < len(sc):
                if k % 2:
                    s.submit_pcm16((sc[t] * 32767).astype("<i2").tobytes())
                else:
                    s.submit(sc[t])
        pool.tick()
        if move is not None:
            move(t, sessions)
    pool.drain()
    return sessions, logs


def test_sharded_pool_places_migrates_and_matches_one_pool():
    cfg = VADConfig(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3,
                    buffer_size=480)
    single, _eng, _calls = make_pool()
    _, want = _run_sessions(single, cfg, 7)
    assert all(sum(1 for e in log if e == "S") == 2 for log in want)

    pool, engines = _sharded(2)
    moved = []

    def move(t, sessions):
        if t == 3:                                   # inside everyone's first utterance: segment open, pre-roll consumed
            for k in (0, 1, 4):
                src = pool.shard_of(sessions[k])
                pool.migrate(sessions[k], 1 - src)
                moved.append((k, src, pool.shard_of(sessions[k])))
        if t == 9:
            pool.migrate(sessions[0], pool.shard_of(sessions[0]))      # to where it already is: nothing happens
    sessions, got = _run_sessions(pool, cfg, 7, move)
    assert got == want                                                  # events, payload bytes and WAV bytes, session by session
    assert [p.session_count for p in pool.shards] in ([4, 3], [3, 4], [5, 2], [2, 5]) and pool.session_count == 7
    assert len(moved) == 3 and all(a != b for _, a, b in moved) and pool.migrations == 3
    assert sum(getattr(e, "restores", 0) for e in engines) == 3
    st = pool.stats()
    assert st["sessions"] == 7 and st["migrations"] == 3 and len(st["shards"]) == 2 and st["frames"] == sum(len(_script(k)) for k in range(7))
    # least-loaded placement and rebalancing after clients left
    for s in [x for x in sessions if pool.shard_of(x) == 0]:
        s.close()
    assert pool.shards[0].session_count == 0
    n1 = pool.shards[1].session_count
    assert pool.shard_of(pool.open_session(cfg)) == 0
    assert pool.rebalance() == (n1 - 1) // 2 and abs(pool.shards[0].session_count - pool.shards[1].session_count) <= 1
    # a session that is closed cannot move; frames submitted by another thread while it moves wait and then land on the new pool
    s = pool.open_session(cfg)
    s.close()
    with pytest.raises(AudioProcessingError, match="closed"):
        pool.migrate(s, 0)
    pool.close()
    assert pool.session_count == 0


def test_frames_submitted_during_a_migration_are_not_lost():
    import threading
    cfg = VADConfig(voice_start_frame_count=1, buffer_size=512)
    pool, engines = _sharded(2)
    s = pool.open_session(cfg, shard=0)
    got = []
    s.set_callbacks(voice_start_callback=lambda: got.append("S"), voice_continue_callback=lambda pcm: got.append("C"))
    loud16 = (np.full(512, 0.5) * 32767).astype("<i2").tobytes()
    stop = threading.Event()
    sent = [0]

    def producer():
        import time
        while not stop.is_set() and sent[0] < 200:       # (at most 256 frames may wait for one stream)
            s.submit_pcm16(loud16)
            sent[0] += 1
            time.sleep(0.0002)

    th = threading.Thread(target=producer)
    th.start()
    for k in range(20):
        pool.tick()
        pool.migrate(s, (k + 1) % 2)
    stop.set()
    th.join()
    pool.drain()
    assert s.frames_done == sent[0] and got[0] == "S" and len(got) == sent[0]      # every frame stepped once: START, then CONTINUEs
    assert pool.migrations == 20 and pool.shard_of(s) == 0
    pool.close()


def test_wire_frames_are_coalesced_into_one_push_per_tick():
    pool, eng, calls = make_pool()
    cfg = VADConfig(voice_start_frame_count=1, buffer_size=512)
    sessions = [pool.open_session(cfg) for _ in range(50)]
    errs = []
    for s in sessions:
        s.set_callbacks(error_callback=errs.append)
    loud16 = (np.full(512, 0.5) * 32767).astype("<i2").tobytes()
    for s in sessions:
        s.submit_pcm16(loud16)
    assert getattr(eng, "push_status_calls", 0) == 0               # nothing has reached the engine yet
    assert pool.tick() == 50 and eng.push_status_calls == 1 and calls == [50]
    # order across the two ingest paths: an odd-sized frame flushes what was collected before it
    a = sessions[0]
    long16 = (np.full(700, 0.5) * 32767).astype("<i2").tobytes()
    a.submit_pcm16(loud16)
    a.submit_pcm16(long16)                                          # longer than the model's frame: pushed directly
    a.submit_pcm16(loud16[:600])
    assert pool.drain() == 3 and a.frames_done == 4
    assert [f.shape for f in eng.frames_seen[-3:]] == [(1, 512)] * 3 and np.all(eng.frames_seen[-1][0, 300:] == 0)
    assert np.all(eng.frames_seen[-2][0] != 0)
    # a frame the engine refuses is reported to its own session only
    for _ in range(300):
        a.submit_pcm16(loud16)
    pool.tick()
    assert len(errs) >= 1 and all("256 frames are waiting" in str(e) for e in errs)
    sessions[1].close()
    pool.close()


def test_asgi_app_over_a_sharded_pool_serves_clients_from_one_conducting_ticker():
    """create_app on a ShardedStreamPool: clients land on different shards, every shard's own ticker thread delivers its clients'
    events, /stats reports per shard, and shutting the app down stops the tickers."""
    from fastapi.testclient import TestClient
    from cutter_vad_amd.server.app import create_app
    pool, engines = _sharded(2)
    app = create_app(pool, tick_interval=0.002)
    loud = (np.full(480, 0.5) * 32767).astype("<i2").tobytes()
    with TestClient(app) as client:
        with client.websocket_connect("/vad?start_frame_count=2&end_frame_count=3") as a, \
                client.websocket_connect("/vad?start_frame_count=2&end_frame_count=3") as b:
            for ws in (a, b):
                assert json.loads(ws.receive_text())["event"] == "INFO"
            assert [p.session_count for p in pool.shards] == [1, 1]              # least-loaded placement
            for ws in (a, b):
                for _ in range(3):
                    ws.send_bytes(loud)
            for ws in (a, b):
                seen = []
                for _ in range(50):
                    seen.append(json.loads(ws.receive_text())["event"])
                    if "VOICE_CONTINUE" in seen:
                        break
                assert seen[:2] == ["VOICE_START", "VOICE_CONTINUE"]
            st = client.get("/stats").json()
            assert st["sessions"] == 2 and len(st["shards"]) == 2 and all(x["frames"] == 3 for x in st["shards"])
        # ONE conducting ticker thread for all shards (round 3 had one per shard: they queued for the interpreter lock)
        assert pool._thread is not None and all(p._thread is None for p in pool.shards)
    assert pool._thread is None and pool.session_count == 0


def test_a_client_that_sends_faster_than_real_time_is_paused_not_refused():
    """The reference processes a frame inside the socket's receive loop, so a client streaming a file at full speed is just read as
    fast as it is processed.  Here a client 96 frames ahead of the pool is not read until the pool has caught up: 900 frames sent
    back to back produce no error (the engine refuses a 257th waiting frame) and are all stepped, in order."""
    from fastapi.testclient import TestClient
    from cutter_vad_amd.server.app import create_app
    pool, eng, calls = make_pool()
    app = create_app(pool, tick_interval=0.001)
    loud = (np.full(480, 0.5) * 32767).astype("<i2").tobytes()
    quiet = np.zeros(480, "<i2").tobytes()
    with TestClient(app) as client:
        with client.websocket_connect("/vad?start_frame_count=2&end_frame_count=3") as ws:
            assert json.loads(ws.receive_text())["event"] == "INFO"
            for i in range(900):
                ws.send_bytes(loud if (i // 10) % 2 == 0 else quiet)             # ten frames of talk, ten of silence
            ws.send_text(json.dumps({"type": "HEARTBEAT"}))
            events = []
            while True:
                m = json.loads(ws.receive_text())
                if m["event"] == "INFO" and m.get("message") == "Heartbeat received":
                    break
                events.append(m["event"])
            assert "ERROR" not in events
            (state,) = app.state.vad["clients"].values()
            deadline = __import__("time").time() + 20
            while state.backlog() > 0 and __import__("time").time() < deadline:
                __import__("time").sleep(0.01)
            assert state.session.frames_done == 900 and state.sent == 900
            # the heartbeat is answered when it is read - after the 900 frames - so by then at most BACKLOG_HIGH + 16 were waiting
            assert events.count("VOICE_START") >= 40
    pool.close()


def test_the_ticker_catches_up_when_a_session_is_ahead_of_real_time():
    """One frame per session and tick; a session that has 100 frames queued (a file, a stall that ended) is not made to wait
    tick_interval between them: a tick that leaves frames staged for the next one is followed by the next one at once."""
    import time
    pool, eng, calls = make_pool()
    pool.tick_interval = 0.05                                   # 100 frames would take 5 s at one per interval
    s = pool.open_session(VADConfig(voice_start_frame_count=1, buffer_size=480))
    for _ in range(100):
        s.submit(LOUD)
    t0 = time.time()
    pool.start()
    while s.frames_done < 100 and time.time() - t0 < 4:
        time.sleep(0.005)
    dt = time.time() - t0
    pool.close()
    assert s.frames_done == 100 and dt < 2.0, dt


def test_voice_continue_as_a_notification_gets_no_payload_and_the_same_calls():
    """The reference server's voice_continue handler ignores its pcm_data (vad_websocket_server.py:420-430): with
    ``continue_payload=False`` the tick calls the callback with b"" at the same frames and builds nothing for it; a reconfigure
    keeps the choice."""
    pool, eng, _ = make_pool()
    cfg = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3, buffer_size=480)
    a, b = pool.open_session(VADConfig(**cfg)), pool.open_session(VADConfig(**cfg))
    la, lb = [], []
    a.set_callbacks(lambda: la.append("S"), lambda wav: la.append("E"), lambda pcm: la.append(len(pcm)))
    b.set_callbacks(lambda: lb.append("S"), lambda wav: lb.append("E"), lambda pcm: lb.append(len(pcm)), continue_payload=False)
    for f in [LOUD] * 4 + [QUIET] * 4:
        a.submit(f)
        b.submit(f)
    pool.drain()
    assert la == ["S", 1920, 1920, 1920, 1920, "E", 1920] and lb == ["S", 0, 0, 0, 0, "E", 0]
    pool.reconfigure(b, VADConfig(**cfg))
    for f in [LOUD] * 3:
        b.submit(f)
    pool.drain()
    assert lb[-2:] == ["S", 0]
    pool.close()


def test_loop_relay_delivers_in_order_with_one_wakeup_per_batch():
    """Events posted from a ticker thread reach the loop in order; a burst costs one call_soon_threadsafe, not one per event; an
    event whose handler raises does not stop the ones behind it."""
    import asyncio
    from cutter_vad_amd.server.app import LoopRelay

    async def main():
        loop = asyncio.get_running_loop()
        relay = LoopRelay(loop)
        wakeups = []
        orig = loop.call_soon_threadsafe
        loop.call_soon_threadsafe = lambda cb, *a: (wakeups.append(1), orig(cb, *a))[1]
        got = []

        def boom():
            raise RuntimeError("handler failed")

        def burst():
            for i in range(1000):
                relay.post(got.append, i)
                if i == 500:
                    relay.post(boom)
        await loop.run_in_executor(None, burst)
        for _ in range(100):
            if len(got) == 1000:
                break
            await asyncio.sleep(0.001)
        assert got == list(range(1000)) and 1 <= len(wakeups) <= 50
    asyncio.run(main())


# ---------------------------------------------------------------------------------------------- ADVICE round 3
def test_a_close_that_raced_a_migration_closes_the_session_where_it_landed():
    """A caller that read ``s.pool`` before a migration finished used to resume on the SOURCE pool with ``s.slot`` already the
    destination's slot number: it popped another client's session there and cancelled / closed that client's stream, and the
    migrated stream was never closed.  Emulated as in the finding: a = (shard 0, slot 0), b = (shard 0, slot 1); a moves to shard 1
    and gets slot number 1 there; the stale ``src.close_session(a)`` must close a on shard 1 and leave b alone."""
    cfg = VADConfig(voice_start_frame_count=1, buffer_size=512)
    pool, engines = _sharded(2)
    filler = pool.open_session(cfg, shard=1)                 # so that a's slot on shard 1 gets b's slot NUMBER on shard 0
    a, b = pool.open_session(cfg, shard=0), pool.open_session(cfg, shard=0)
    src = a.pool
    assert (a.slot, b.slot, filler.slot) == (0, 1, 0)
    pool.migrate(a, 1)
    assert a.pool is pool.shards[1] and a.slot == 1 and a._home == (pool.shards[1], 1)
    src.close_session(a)                                     # the stale call
    assert a.closed and not b.closed
    assert pool.shards[0]._by_slot[1] is b and pool.shards[0]._sessions == {1: b}
    assert pool.shards[1]._sessions == {0: filler} and 1 in engines[1].closed and engines[0].closed == [0]
    b.submit(LOUD)
    pool.tick()
    assert b.frames_done == 1
    pool.close()


def test_close_and_reconfigure_wait_for_a_running_migration():
    """Threaded: a migration is held between the drain and the state transfer; close_session (through the sharded front and
    directly on the session) and reconfigure started meanwhile act on the pool the session ends up on."""
    import threading
    import time
    cfg = VADConfig(voice_start_frame_count=1, buffer_size=512)
    for how in ("front", "direct", "reconfigure"):
        pool, engines = _sharded(2)
        a, b = pool.open_session(cfg, shard=0), pool.open_session(cfg, shard=0)
        pool.open_session(cfg, shard=1)
        entered, release = threading.Event(), threading.Event()
        orig = engines[0].save_stream

        def slow_save(slot, orig=orig):
            entered.set()
            release.wait(5)
            return orig(slot)
        engines[0].save_stream = slow_save
        mover = threading.Thread(target=lambda: pool.migrate(a, 1))
        mover.start()
        assert entered.wait(5)
        new_cfg = VADConfig(voice_start_frame_count=3, buffer_size=512)
        act = {"front": lambda: pool.close_session(a), "direct": a.close, "reconfigure": lambda: pool.reconfigure(a, new_cfg)}[how]
        other = threading.Thread(target=act)
        other.start()
        time.sleep(0.05)
        assert other.is_alive() and not a.closed                 # it waits: the session is between two shards
        release.set()
        mover.join(5)
        other.join(5)
        assert not mover.is_alive() and not other.is_alive()
        assert a.pool is pool.shards[1] and not b.closed and pool.shards[0]._sessions == {b.slot: b}
        if how == "reconfigure":
            assert not a.closed and a.config is new_cfg and engines[1].thr[a.slot][4] == 3
        else:
            assert a.closed and a.slot in engines[1].closed and a.slot not in pool.shards[1]._sessions
        pool.close()


def test_frames_that_arrive_while_a_session_moves_are_held_not_waited_for():
    """``submit`` on a moving session used to spin with time.sleep - on the ASGI app that is the event loop's thread, so one
    migration froze every socket of the loop.  Now the frame is held on the session and replayed, in order, on the pool it lands on."""
    import threading
    import time
    cfg = VADConfig(voice_start_frame_count=1, buffer_size=512)
    pool, engines = _sharded(2)
    s = pool.open_session(cfg, shard=0)
    got = []
    s.set_callbacks(voice_start_callback=lambda: got.append("S"), voice_continue_callback=lambda pcm: got.append(len(pcm)))
    entered, release = threading.Event(), threading.Event()
    orig = engines[0].save_stream

    def slow_save(slot):
        entered.set()
        release.wait(5)
        return orig(slot)
    engines[0].save_stream = slow_save
    mover = threading.Thread(target=lambda: pool.migrate(s, 1))
    mover.start()
    assert entered.wait(5) and s.moving
    t0 = time.perf_counter()
    s.submit(LOUD)                                           # float32 frame, 480 samples
    s.submit_pcm16((np.full(512, 0.5) * 32767).astype("<i2").tobytes())
    s.submit(LOUD)
    assert time.perf_counter() - t0 < 0.05 and len(s._held) == 3 and s.moving       # returned at once, nothing reached an engine
    release.set()
    mover.join(5)
    assert not s.moving and len(s._held) == 0 and pool.shard_of(s) == 1
    pool.drain()
    assert s.frames_done == 3 and got == ["S", 4 * 512, 4 * 480]      # START on the first, CONTINUE payloads of the next two, in order
    pool.close()


class _FailingTicks:
    """tick_run raises for the first `n` calls, dropping one queued frame per stream like the engine does, and says whose"""

    def __init__(self, eng, n):
        self.eng, self.left, self.orig = eng, n, eng.tick_run

    def __call__(self, denoise=0.01):
        if self.left > 0:
            self.left -= 1
            q = self.eng.__dict__.setdefault("_tickq", {})
            slots = [s for s in list(q) if q[s]]
            lens = [int(q[s].pop(0)[1].size) for s in slots]
            for s in slots:
                if not q[s]:
                    del q[s]
            self.eng.last_tick_lost = (np.array(slots, np.int64), np.array(lens, np.int32))
            self.eng.last_tick_staged_next = sum(len(v) for v in q.values())
            raise RuntimeError("device fault (scripted)")
        return self.orig(denoise)


def test_backlog_counts_lost_frames_and_a_failed_tick_waits_its_interval():
    pool, eng, _ = make_pool()
    s = pool.open_session(VADConfig(voice_start_frame_count=1, buffer_size=480))
    errors = []
    s.set_callbacks(error_callback=errors.append)
    eng.tick_run = _FailingTicks(eng, 100)
    for _ in range(120):
        s.submit(LOUD)
    pool.backlog = 7                                         # left over by an earlier successful tick
    for k in range(100):
        assert pool.tick() == 0
        assert pool.backlog == 0                             # the ticker does not hammer a failing engine back to back
    assert s.lost == 100 and len(errors) == 100 and s.frames_done == 0
    pool.drain()
    assert s.frames_done == 20 and s.lost == 100             # sent 120 = stepped 20 + lost 100: the app's backlog is 0 again
    pool.close()


def test_a_paused_socket_is_read_again_after_a_failing_engine():
    """`sent - frames_done` stayed high for ever after ~97 lost frames and the receive loop never read the socket again."""
    import asyncio
    from cutter_vad_amd.server import app as appmod
    pool, eng, _ = make_pool()
    s = pool.open_session(VADConfig(voice_start_frame_count=1, buffer_size=480))

    class Client:
        backlog = appmod.ClientSession.backlog
        wait_for_pool = appmod.ClientSession.wait_for_pool
        session, session_error, sent = s, None, 0
    c = Client()
    eng.tick_run = _FailingTicks(eng, 150)
    for _ in range(150):
        s.submit(LOUD)
        c.sent += 1
    for _ in range(150):
        pool.tick()
    assert s.lost == 150 and c.backlog() == 0                # accounted for: nothing to wait for
    asyncio.run(asyncio.wait_for(c.wait_for_pool(0.001), 2))
    # frames lost WITHOUT the engine saying whose (an old count, a bug): the wait is bounded by the pool's progress
    c.sent += 200
    t0 = __import__("time").time()
    asyncio.run(asyncio.wait_for(c.wait_for_pool(0.001), 5))
    assert c.backlog() == 0 and __import__("time").time() - t0 < 3
    # a closed session ends the wait at once
    c.sent += 200
    s.close()
    asyncio.run(asyncio.wait_for(c.wait_for_pool(0.001), 1))
    pool.close()


def test_loop_relay_does_not_keep_its_event_loop_alive():
    import asyncio
    import gc
    import weakref
    from cutter_vad_amd.server.app import _RELAYS, relay_of
    loop = asyncio.new_event_loop()
    r = relay_of(loop)
    assert r.loop is loop and relay_of(loop) is r
    wl, wr = weakref.ref(loop), weakref.ref(r)
    loop.close()
    del loop, r
    gc.collect()
    assert wl() is None and wr() is None and all(k is not None for k in _RELAYS.keys())

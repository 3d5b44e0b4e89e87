#!/usr/bin/env python3
"""Host-inclusive throughput of the shared stream pool: N sessions, one 30 ms int16 frame each per
tick, as the websocket server receives them (int16 bytes -> submit_pcm16), then ONE tick = one launch
(H2D + kernel + D2H) + vectorised fan-out.  Prints one JSON object per N.  Real time needs 33.3 ticks/s."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutter_vad_amd import VADConfig  # noqa: E402
from cutter_vad_amd.server import ShardedStreamPool, SharedStreamPool  # noqa: E402
from tests.signals import make_streams  # noqa: E402


DEVICES = None          # --devices 0,1,...: one pool per listed GPU behind a ShardedStreamPool (the same ordinal twice = two engines on one GPU)


def make_pool(**kw):
    if DEVICES is None:
        return SharedStreamPool(max_streams=8192, **kw)
    return ShardedStreamPool(devices=DEVICES, max_streams=8192, **kw)


def tick_us(pool):
    engines = [p.engine for p in pool.shards] if hasattr(pool, "shards") else [pool.engine]
    return [max(e.last_tick_us[k] for e in engines) for k in range(3)]


CONTINUE = None         # "payload" / "notify": also register a voice_continue callback (the reference server does, and ignores its bytes)


def run(n, ticks=40):
    pool = make_pool()
    cfg = VADConfig(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6,
                    voice_end_frame_count=12, buffer_size=480)
    sessions = [pool.open_session(cfg) for _ in range(n)]
    counts = {"start": 0, "end": 0}
    counts["continue"] = 0
    for s in sessions:
        if CONTINUE is None:
            s.set_callbacks(lambda: counts.__setitem__("start", counts["start"] + 1),
                            lambda wav: counts.__setitem__("end", counts["end"] + 1), None)
        else:
            s.set_callbacks(lambda: counts.__setitem__("start", counts["start"] + 1),
                            lambda wav: counts.__setitem__("end", counts["end"] + 1),
                            lambda pcm: counts.__setitem__("continue", counts["continue"] + 1), continue_payload=CONTINUE == "payload")
    x = make_streams(min(n, 512), ticks + 5, seed=9)[:, :, :480]
    pcm = np.clip(x * 32767.0, -32768, 32767).astype("<i2")
    wire = [[pcm[k % pcm.shape[0], t].tobytes() for k in range(n)] for t in range(ticks + 5)]
    t_sub = t_tick = 0.0
    c_us = [0.0, 0.0, 0.0]
    for t in range(ticks + 5):
        a = time.perf_counter()
        for k, s in enumerate(sessions):
            s.submit_pcm16(wire[t][k])
        b = time.perf_counter()
        pool.tick()
        c = time.perf_counter()
        if t >= 5:
            t_sub += b - a
            t_tick += c - b
            for k, v in enumerate(tick_us(pool)):
                c_us[k] += v
    c_inbox = all(p._wire is not None for p in (pool.shards if hasattr(pool, "shards") else [pool]))
    pool.close()
    return {"sessions": n, "ticks": ticks, "voice_continue": CONTINUE or "no callback", "ingest": "submit_pcm16 per session (the ASGI app's path): frames collect in the pool's inbox, " +
            ("the C one (_wirebox): one vad_tick_push_gather per tick" if c_inbox else "the Python one (VAD_POOL_WIREBOX=0): one join + vad_tick_push_status per tick"),
            "devices": DEVICES, "decode_submit_ms_per_tick": t_sub / ticks * 1e3,
            "tick_in_C_us": {"swap": c_us[0] / ticks, "gpu": c_us[1] / ticks, "segments": c_us[2] / ticks},
            "tick_ms": t_tick / ticks * 1e3, "frames_per_s_host_inclusive": n * ticks / (t_sub + t_tick),
            "real_time_factor": (n * ticks / (t_sub + t_tick)) / (n * 1000 / 30), "events": counts}


def run_batched(n, ticks=40):
    """The same load through the batch ingest a native front end would use (vad_tick_push_many): one call hands over the tick's
    frames of all sockets; everything else - tick, events, WAV payloads on END - is the same code path."""
    pool = SharedStreamPool(max_streams=8192)           # (one device: the batch call addresses one engine's slots)
    cfg = VADConfig(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6,
                    voice_end_frame_count=12, buffer_size=480)
    sessions = [pool.open_session(cfg) for _ in range(n)]
    counts = {"start": 0, "end": 0}
    for s in sessions:
        s.set_callbacks(lambda: counts.__setitem__("start", counts["start"] + 1),
                        lambda wav: counts.__setitem__("end", counts["end"] + 1), None)
    slots = np.array([s.slot for s in sessions], np.int64)
    x = make_streams(min(n, 512), ticks + 5, seed=9)[:, :, :480]
    pcm = np.clip(x * 32767.0, -32768, 32767).astype("<i2")
    wire = [np.ascontiguousarray(pcm[np.arange(n) % pcm.shape[0], t]) for t in range(ticks + 5)]
    t_sub = t_tick = 0.0
    c_us = [0.0, 0.0, 0.0]
    for t in range(ticks + 5):
        a = time.perf_counter()
        pool.engine.tick_push_many(slots, wire[t], gate_on=True)
        b = time.perf_counter()
        pool.tick()
        c = time.perf_counter()
        if t >= 5:
            t_sub += b - a
            t_tick += c - b
            for k in range(3):
                c_us[k] += pool.engine.last_tick_us[k]
    pool.close()
    return {"sessions": n, "ingest": "vad_tick_push_many", "tick_in_C_us": {"swap": c_us[0] / ticks, "gpu": c_us[1] / ticks, "segments": c_us[2] / ticks}, "submit_ms_per_tick": t_sub / ticks * 1e3, "tick_ms": t_tick / ticks * 1e3,
            "frames_per_s_host_inclusive": n * ticks / (t_sub + t_tick),
            "real_time_factor": (n * ticks / (t_sub + t_tick)) / (n * 1000 / 30), "events": counts}


def run_rates(n, ticks=40):
    """n sessions, a quarter each at 8 / 16 / 24 / 48 kHz (SharedStreamPool(convert_rates=True)): every tick carries 32 ms int16
    chunks at the client's rate, resampled on the GPU inside the tick (vad_tick_push_rate)."""
    pool = SharedStreamPool(max_streams=8192, convert_rates=True)
    rates = (8000, 16000, 24000, 48000)
    sessions, counts = [], {"start": 0, "end": 0}
    for k in range(n):
        sr = rates[k % 4]
        s = pool.open_session(VADConfig(sample_rate=sr, buffer_size=512 * sr // 16000, vad_start_probability=0.4,
                                        vad_end_probability=0.3, voice_start_frame_count=6, voice_end_frame_count=12))
        s.set_callbacks(lambda: counts.__setitem__("start", counts["start"] + 1),
                        lambda wav: counts.__setitem__("end", counts["end"] + 1), None)
        sessions.append(s)
    x = make_streams(64, 3 * (ticks + 5), seed=9).reshape(64, -1)
    pcm = np.clip(x * 32767.0, -32768, 32767).astype("<i2")
    wire = []
    for t in range(ticks + 5):
        row = []
        for k in range(n):
            L = 512 * rates[k % 4] // 16000
            row.append(pcm[k % 64, t * L:(t + 1) * L].tobytes())
        wire.append(row)
    t_sub = t_tick = 0.0
    c_us = [0.0, 0.0, 0.0]
    for t in range(ticks + 5):
        a = time.perf_counter()
        for k, s in enumerate(sessions):
            s.submit_pcm16(wire[t][k])
        b = time.perf_counter()
        pool.tick()
        c = time.perf_counter()
        if t >= 5:
            t_sub += b - a
            t_tick += c - b
            for k in range(3):
                c_us[k] += pool.engine.last_tick_us[k]
    st = pool.stats()
    pool.close()
    return {"sessions": n, "ingest": "submit_pcm16, a quarter each at 8 / 16 / 24 / 48 kHz (32 ms chunks)",
            "tick_in_C_us": {"swap": c_us[0] / ticks, "gpu": c_us[1] / ticks, "segments": c_us[2] / ticks},
            "submit_ms_per_tick": t_sub / ticks * 1e3, "tick_ms": t_tick / ticks * 1e3, "launches_per_tick": st["launches"] / st["ticks"],
            "frames_per_s_host_inclusive": n * ticks / (t_sub + t_tick),
            "real_time_factor": (n * ticks / (t_sub + t_tick)) / (n * 1000 / 32), "events": counts}


def run_fake_shards(n_shards, sessions, ticks=60, talking=0.5):
    """Host-side rehearsal of the north-star size (65 536 sessions = 8 GPUs x 8 192) WITHOUT GPUs: `n_shards` engines of the REAL
    host code (csrc/engine.cpp: tick assembler, segment arena, the C ABI) over the HIP stand-in of tools/san_tick/ (tests/standin.py:
    host stand-ins for the kernel launches, p = |first sample|), behind ShardedStreamPool - one pool, one ticker thread per shard -
    with the C inbox and the reference server's callbacks (voice_continue as a notification).  What it measures is everything the
    serving process does on the host per 30 ms of audio of every client: `submit_pcm16` per session on ONE thread (the event loop's
    role), then one conducted tick of all shards (`ShardedStreamPool.tick`): the engines' share side by side on C threads behind one
    release of the interpreter lock, then the events fanned out shard by shard on the conducting thread.  Half the sessions talk (their frames open and close
    segments: START / CONTINUE per frame / END with a WAV payload), half are silent."""
    import threading
    from cutter_vad_amd import _ffi, weights_io
    from cutter_vad_amd.pool import EnginePool
    from tests import standin
    _ffi.LIB_PATH = standin.build(os.path.join(os.path.dirname(os.path.abspath(__file__)), "san_tick", "_build"))
    per = sessions // n_shards
    pool = ShardedStreamPool(shards=[SharedStreamPool(max_streams=per, pool=EnginePool(), device_id=0) for _ in range(n_shards)])
    assert all(p._wire is not None and p._wire_entry is not None for p in pool.shards), "the C inbox is part of what is measured"
    cfg = VADConfig(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6, voice_end_frame_count=12, buffer_size=480)
    counts = {"start": 0, "end": 0, "continue": 0, "wav_bytes": 0}
    lock = threading.Lock()                     # callbacks run on eight ticker threads

    def on_start():
        with lock:
            counts["start"] += 1

    def on_end(wav):
        with lock:
            counts["end"] += 1
            counts["wav_bytes"] += len(wav)

    def on_cont(pcm):
        counts["continue"] += 1                 # (unlocked on purpose: the reference's handler only formats a message)
    t0 = time.perf_counter()
    sess = []
    for k in range(sessions):
        s = pool.open_session(cfg, shard=k % n_shards)
        s.set_callbacks(on_start, on_end, on_cont, continue_payload=False)
        sess.append(s)
    t_open = time.perf_counter() - t0
    # the stand-in model's probability is |first sample|
    loud = np.zeros(480, "<i2"); loud[0] = int(0.9 * 32767); loud[1:] = (np.random.default_rng(1).standard_normal(479) * 3000).astype("<i2")
    quiet = np.zeros(480, "<i2")
    loud_b, quiet_b = loud.tobytes(), quiet.tobytes()
    n_talk = int(sessions * talking)
    tick_wall = [[] for _ in range(n_shards)]   # per shard: the fan-out of its events on the conducting thread
    for k, p in enumerate(pool.shards):
        orig = p._fan_out

        def timed(res, wavs, orig=orig, k=k):
            a = time.perf_counter()
            n = orig(res, wavs)
            tick_wall[k].append(time.perf_counter() - a)
            return n
        p._fan_out = timed
    t_sub = t_tick = 0.0
    warm = 5
    # a talker's 40-tick cycle (20 loud, 20 quiet) starts at its own phase, so STARTs and ENDs (a WAV payload each) are spread
    # over the ticks as they are among real clients, instead of 32 768 segments ending in one tick
    # (the phase is the session's index WITHIN its shard - sessions are dealt to the shards round robin - so that every shard sees
    # the same share of STARTs and ENDs in every tick; a phase of (t + k) % 40 puts all of a tick's ENDs on one shard)
    rows = [[(loud_b if ((t + k // n_shards) % 40) < 20 else quiet_b) if k < n_talk else quiet_b for k in range(sessions)] for t in range(40)]
    for t in range(ticks + warm):
        row = rows[t % 40]
        a = time.perf_counter()
        for s, f in zip(sess, row):
            s.submit_pcm16(f)
        b = time.perf_counter()
        n = pool.tick()
        c = time.perf_counter()
        assert n == sessions, n
        if t >= warm:
            t_sub += b - a
            t_tick += c - b
        else:
            for w in tick_wall:
                w.clear()
    per_shard = [{"mean_ms": float(np.mean(w)) * 1e3, "max_ms": float(np.max(w)) * 1e3} for w in tick_wall]
    fan = sum(x["mean_ms"] for x in per_shard)
    stats = pool.stats()
    pool.close()
    per_round = (t_sub + t_tick) / ticks
    return {"mode": "fake shards (real engine.cpp + _wirebox + pools; kernels = host stand-ins, tools/san_tick)", "shards": n_shards,
            "sessions": sessions, "sessions_per_shard": per, "talking_fraction": talking, "ticks": ticks, "host_cores": os.cpu_count(),
            "open_all_sessions_s": t_open, "submit_ms_per_round": t_sub / ticks * 1e3, "tick_all_shards_ms_per_round": t_tick / ticks * 1e3,
            "engines_side_by_side_ms_per_round": t_tick / ticks * 1e3 - fan, "fan_out_all_shards_ms_per_round": fan,
            "per_shard_fan_out_wall": per_shard, "round_ms": per_round * 1e3, "audio_ms_per_round": 30.0,
            "real_time_factor": 0.030 / per_round, "frames_per_s_host_inclusive": sessions / per_round,
            "frames_per_launch": stats["frames_per_launch"], "events": counts}


def warm_arena():
    """The engine's segment arena grows by 8 MB chunks and first-touches them inside the tick (≈ 0.15 ms of page faults per tick at
    8 192 talking sessions while it grows); closed sessions hand their blocks back, so a server that has been up for a minute runs
    on touched memory.  One untimed pass at full size puts the process in that state - the lines below are steady-state ticks."""
    if DEVICES is None:
        run_batched(8192)


if __name__ == "__main__":
    if "--fake-shards" in sys.argv:
        sys.argv.append("--cold")                  # no GPU in this mode
    if "--cold" not in sys.argv:
        warm_arena()
    else:
        sys.argv.remove("--cold")
    if "--fake-shards" in sys.argv:
        i = sys.argv.index("--fake-shards")
        shards = int(sys.argv[i + 1])
        n = int(sys.argv[sys.argv.index("--sessions") + 1]) if "--sessions" in sys.argv else 8192 * shards
        print(json.dumps(run_fake_shards(shards, n)), flush=True)
        sys.exit(0)
    if "--devices" in sys.argv:
        i = sys.argv.index("--devices")
        DEVICES = [int(d) for d in sys.argv[i + 1].split(",")]
        del sys.argv[i:i + 2]
        for n in (2048, 8192):
            print(json.dumps(run(n)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "continue":          # every talking session gets a voice_continue call per tick
        for mode in ("payload", "notify"):
            CONTINUE = mode
            print(json.dumps(run(8192)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rates":
        for n in (2048, 8192):
            print(json.dumps(run_rates(n)), flush=True)
        sys.exit(0)
    for n in (256, 2048, 8192):
        print(json.dumps(run(n)), flush=True)
    print(json.dumps(run_batched(8192)), flush=True)

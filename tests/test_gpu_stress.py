"""Randomised end-to-end stress of the C ABI on the GPU: open / close / reset / step / step_multi / save / restore in
random order, random slot subsets and call sizes (both staging paths of the host-pointer entry points), both frame
formats - every probability checked against a per-stream oracle replay."""

import numpy as np
import pytest

from cutter_vad_amd import weights_io
from tests.signals import make_streams

pytestmark = pytest.mark.gpu


import os

# VAD_STRESS_SEEDS=N runs N differently seeded sequences per model (soak runs; the default suite runs one)
_SEEDS = list(range(int(os.environ.get("VAD_STRESS_SEEDS", "1"))))


@pytest.mark.parametrize("seed", _SEEDS)
@pytest.mark.parametrize("version", [5, 4])
def test_random_api_sequence_matches_per_stream_oracle(version, seed):
    from cutter_vad_amd import VADError
    from cutter_vad_amd.engine import Engine
    from oracle import oracle
    with open(weights_io.packaged_blob_path(version), "rb") as f:
        blob = f.read()
    om = oracle.OracleModel(blob, "f64")
    rng = np.random.default_rng(1000 + version + 17 * seed)
    POOL = 300
    audio = make_streams(POOL, 64, seed=31 + version + 5 * seed)            # every logical stream has 64 frames to play
    # Every call is checked LOCALLY: the oracle starts from the device's own state before the call (vad_stream_get_state
    # after the previous one), so a deviation is attributed to the call that made it and is not carried on by the LSTM.
    # Fixed bars for both models (V4's two real STFT bins are summed in float64: tests/test_gpu_v4.py, DESIGN "Numerics").
    tol = 2e-5 if version == 5 else 3e-5
    with Engine(blob, model_version=version, max_streams=POOL) as eng:
        slot_of, pos, state, saved = {}, {}, {}, {}
        worst, worst_state, checked = 0.0, 0.0, 0
        for it in range(260):
            op = rng.choice(["open", "step", "step", "step", "multi", "close", "reset", "save", "restore"])
            live = sorted(slot_of)
            if op == "open" or len(live) < 8:
                free = [i for i in range(POOL) if i not in slot_of]
                for i in rng.choice(free, size=min(len(free), int(rng.integers(1, 40))), replace=False):
                    slot_of[int(i)] = int(eng.open_stream())
                    pos[int(i)], state[int(i)] = 0, np.zeros((1, 256), np.float32)
                    assert not eng.get_state(slot_of[int(i)]).any()
            elif op in ("step", "multi"):
                k = int(rng.integers(1, len(live) + 1))
                ids = [int(i) for i in rng.choice(live, size=k, replace=False)]
                T = 1 if op == "step" else int(rng.integers(2, 4))
                ids = [i for i in ids if pos[i] + T <= 64]
                if not ids:
                    continue
                fr = np.stack([audio[i, pos[i]:pos[i] + T] for i in ids])                      # [k, T, 512]
                use_i16 = bool(rng.integers(0, 2))
                if use_i16:
                    q = np.clip(np.round(fr * 32767.0), -32768, 32767).astype(np.int16)
                    fr_ref = (q.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
                    send = q
                else:
                    fr_ref, send = fr, fr
                slots = [slot_of[i] for i in ids]
                if T == 1:
                    got = eng.step(slots, send[:, 0])[:, None]
                else:
                    got, _ = eng.step_multi(slots, send)
                for r, i in enumerate(ids):
                    st = state[i].copy()
                    for t in range(T):
                        ref = om.step_batch(oracle.denoise(fr_ref[r, t]).reshape(1, 512), st, nthreads=1)[0]
                        worst = max(worst, abs(float(got[r, t]) - float(ref)))
                        checked += 1
                    pos[i] += T
                    dev = eng.get_state(slot_of[i]).reshape(1, 256)
                    dstate = float(np.abs(dev - st).max())
                    worst_state = max(worst_state, dstate)
                    state[i] = dev.copy()                 # the next call's oracle starts where the device is
            elif op == "close" and len(live) > 8:
                for i in rng.choice(live, size=int(rng.integers(1, 6)), replace=False):
                    eng.close_stream(slot_of.pop(int(i)))
                    saved.pop(int(i), None)
            elif op == "reset":
                ids = [int(i) for i in rng.choice(live, size=int(rng.integers(1, 5)), replace=False)]
                eng.reset([slot_of[i] for i in ids])
                for i in ids:
                    state[i] = np.zeros((1, 256), np.float32)
                    assert not eng.get_state(slot_of[i]).any()
            elif op == "save":
                i = int(rng.choice(live))
                saved[i] = (eng.save_stream(slot_of[i]), pos[i], state[i].copy())
            elif op == "restore" and saved:
                i = int(rng.choice(sorted(saved)))
                blob_i, p_i, st_i = saved[i]
                eng.restore_stream(slot_of[i], blob_i)
                pos[i], state[i] = p_i, st_i.copy()
                assert np.array_equal(eng.get_state(slot_of[i]).reshape(1, 256), st_i)      # the blob carries (h, c) bit for bit
        assert checked > 1500 and worst <= tol and worst_state <= 2e-4, (checked, worst, worst_state)
        # bookkeeping survived: a slot cannot be stepped twice in one call, closed slots are refused
        s = slot_of[sorted(slot_of)[0]]
        with pytest.raises(Exception):
            eng.step([s, s], np.zeros((2, 512), np.float32))
        last = sorted(slot_of)[-1]
        gone = slot_of.pop(last)
        eng.close_stream(gone)
        with pytest.raises((VADError, Exception)):
            eng.step([gone], np.zeros((1, 512), np.float32))
        assert eng.info()["open_streams"] == len(slot_of)


@pytest.mark.parametrize("version", [5, 4])
def test_concurrent_callers_on_one_engine_match_a_serial_run(version):
    """SURVEY §8b threading: the entry points are thread-safe per engine, a slot is stepped by one caller at a time.
    Six threads step disjoint slot groups of ONE engine (host-pointer API, both staging paths by group size) while a
    seventh keeps resampling on the same engine; every group must reproduce, bit for bit, what a serial run produces."""
    import threading
    from cutter_vad_amd.engine import Engine
    with open(weights_io.packaged_blob_path(version), "rb") as f:
        blob = f.read()
    sizes = [1, 7, 32, 33, 200, 700]
    T = 12
    audio = [make_streams(n, T, seed=50 + i) for i, n in enumerate(sizes)]
    with Engine(blob, model_version=version, max_streams=sum(sizes)) as eng:
        groups = [eng.open_streams(n) for n in sizes]
        serial = [np.stack([eng.step(g, a[:, t]) for t in range(T)], axis=1) for g, a in zip(groups, audio)]
        states = [eng.get_state(int(g[-1])) for g in groups]
        for g in groups:
            eng.reset(g)
        got = [None] * len(sizes)
        errors = []
        stop = threading.Event()

        def worker(i):
            try:
                got[i] = np.stack([eng.step(groups[i], audio[i][:, t]) for t in range(T)], axis=1)
            except Exception as e:  # pragma: no cover
                errors.append(e)

        def resampler():
            x = (0.1 * np.random.default_rng(9).standard_normal((40, 768))).astype(np.float32)
            first = eng.resample(x, 24000)
            try:
                while not stop.is_set():
                    assert np.array_equal(eng.resample(x, 24000), first)
            except Exception as e:  # pragma: no cover
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(sizes))]
        side = threading.Thread(target=resampler)
        side.start()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        stop.set()
        side.join()
        assert not errors, errors
        for i in range(len(sizes)):
            assert np.array_equal(got[i], serial[i]), sizes[i]
            assert np.array_equal(eng.get_state(int(groups[i][-1])), states[i])


def test_producers_push_while_the_tick_runs_and_every_stream_sees_its_frames_in_order():
    """The serving tick under load, on the real kernels (the CPU twin with ThreadSanitizer / AddressSanitizer is
    tests/test_tick_sanitizers.py): four producer threads push frames (float32, int16 wire frames, 48 kHz chunks resampled inside
    the tick, over-long frames) for their own streams WHILE another thread loops vad_tick_run and takes finished segments; a
    fifth thread opens / closes streams with frames still queued.  Per stream the result must be what a serial run gives: the
    same probabilities in the same order (bit for bit - a frame is stepped exactly once, in push order), and segments of exactly
    the samples the serial run keeps."""
    import threading
    import time
    from cutter_vad_amd.engine import Engine
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    import os
    P, PER, K = 4, 48, 48
    n = P * PER
    pcm = np.load(os.path.join(os.path.dirname(__file__), "golden", "speech16k_i16.npz"))["pcm"].astype(np.float32) / np.float32(32767.0)
    span = K * 512
    # every stream hears the reference's speech clip from its own offset: talk and pauses, so segments open AND finish
    base = np.stack([pcm[(i * 3001) % (pcm.size - span):][:span].reshape(K, 512) for i in range(n)]).astype(np.float32)
    rng = np.random.default_rng(3)
    chunks48 = np.repeat(base, 3, axis=2) * np.float32(0.9)                 # 1536-sample chunks (a crude 48 kHz rendition: any signal will do)
    kind = np.arange(n) % 4                                                 # 0 f32 | 1 int16 | 2 48 kHz chunk | 3 over-long f32 (600)
    tail = (0.05 * rng.standard_normal((n, K, 88))).astype(np.float32)
    thr = (0.4, 0.3, 0.5, 0.5, 2, 4)

    def frame(i, k):
        if kind[i] == 1:
            return np.clip(np.round(base[i, k] * 32767.0), -32768, 32767).astype(np.int16), None
        if kind[i] == 2:
            return chunks48[i, k], 48000
        if kind[i] == 3:
            return np.concatenate([base[i, k], tail[i, k]]), None
        return base[i, k], None

    def run(eng, concurrent):
        eng.tick_enable_segments(True)
        slots = eng.open_streams(n)
        eng.set_thresholds_many(slots, thr)
        probs = {int(s): [] for s in slots}
        segs = {int(s): [] for s in slots}
        mine = set(int(s) for s in slots)
        left = [P + (1 if concurrent else 0)]
        lock = threading.Lock()
        errors = []

        def producer(t):
            try:
                for k in range(K):
                    for j in range(PER):
                        i = t * PER + j
                        x, sr = frame(i, k)
                        eng.tick_push(int(slots[i]), x, True, sample_rate=sr)
                    if concurrent:
                        time.sleep(0.0005)
            except Exception as e:  # pragma: no cover
                errors.append(e)
            finally:
                with lock:
                    left[0] -= 1

        def churn():
            try:
                for r in range(60):
                    s = eng.open_streams(4)
                    for k in range(3):
                        for q in s:
                            eng.tick_push(int(q), base[0, k], True)
                    if r % 2:
                        for q in s:
                            eng.tick_cancel(int(q))
                    for q in s:
                        eng.close_stream(int(q))
            except Exception as e:  # pragma: no cover
                errors.append(e)
            finally:
                with lock:
                    left[0] -= 1

        ths = [threading.Thread(target=producer, args=(t,)) for t in range(P)]
        if concurrent:
            ths.append(threading.Thread(target=churn))
            for t in ths:
                t.start()
        else:
            for t in ths:
                t.run()
        while True:
            done = left[0] == 0
            sl, p, ev, _seg, _gs, _fr, _ns = eng.tick_run(0.01)
            for s, pv, e in zip(sl.tolist(), p.tolist(), ev.tolist()):
                if s in mine:
                    probs[s].append(pv)
                    if e & 2:
                        segs[s].append(eng.tick_take_segment(s))
            if done and sl.size == 0:
                break
        for t in ths:
            if concurrent:
                t.join()
        assert not errors, errors
        return [np.array(probs[int(s)], np.float32) for s in slots], [segs[int(s)] for s in slots]

    with Engine(blob, max_streams=1024) as eng:
        serial_p, serial_s = run(eng, False)
    with Engine(blob, max_streams=1024) as eng:
        conc_p, conc_s = run(eng, True)
    n_seg = 0
    for i in range(n):
        assert serial_p[i].size == K and np.array_equal(serial_p[i], conc_p[i]), i
        assert len(serial_s[i]) == len(conc_s[i]), i
        for a, b in zip(serial_s[i], conc_s[i]):
            assert np.array_equal(a, b), i
            n_seg += 1
    assert n_seg >= 20                                                      # segments do finish in this scenario

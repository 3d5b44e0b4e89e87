"""GPU tests of the drop-in surface: VADWrapper / VADProcessor / StreamBatch over the HIP engine."""

import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def speech():
    return np.load(os.path.join(GOLD, "speech16k_i16.npz"))["pcm"]


def test_device_state_machine_replays_reference_traces():
    """Every scenario recorded from the reference's VADProcessor, replayed through the device-side
    state machine (probabilities pass through float32, as they do from the model)."""
    from cutter_vad_amd import weights_io
    from cutter_vad_amd.engine import Engine
    from oracle import oracle
    with open(os.path.join(GOLD, "state_machine.json")) as f:
        scen = json.load(f)["scenarios"]
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    with Engine(blob, max_streams=64) as eng:
        for name, s in scen.items():
            c = s["config"]
            thr = (c.get("vad_start_probability", 0.7), c.get("vad_end_probability", 0.7),
                   c.get("voice_start_ratio", 0.8), c.get("voice_end_ratio", 0.95),
                   c.get("voice_start_frame_count", 10), c.get("voice_end_frame_count", 50))
            slot = eng.open_stream()
            eng.set_thresholds(slot, *thr)
            p32 = np.asarray(s["probs"], np.float32)
            ev, seg = eng.debug_sm_replay(slot, p32)
            sm = oracle.StateMachine(*thr)
            ref = [sm.step(float(p), 1) for p in p32]
            assert list(ev) == [r[0] for r in ref], name
            assert list(seg) == [r[1] if r[0] & 2 else 0 for r in ref], name
            if all(float(np.float32(p)) == p or abs(p - thr[0]) > 1e-6 and abs(p - thr[1]) > 1e-6 for p in s["probs"]):
                assert list(ev) == s["events"], name          # identical to the reference trace itself
            eng.close_stream(slot)


def test_vadwrapper_end_to_end_four_segments(speech):
    """The reference's only real-model expectation (examples/test_python_vad_client.py:200-223),
    through the real product path: VADWrapper -> VADProcessor -> C ABI -> HIP kernel."""
    from cutter_vad_amd import SampleRate, SileroModelVersion, VADConfig, VADWrapper
    g = np.load(os.path.join(GOLD, "e2e_client_config.npz"))
    cfg = VADConfig(sample_rate=SampleRate(16000), model_version=SileroModelVersion.V5, vad_start_probability=0.4,
                    vad_end_probability=0.3, voice_start_ratio=0.8, voice_end_ratio=0.95, voice_start_frame_count=6,
                    voice_end_frame_count=12, enable_denoising=True, auto_convert_sample_rate=True, buffer_size=480)
    log, wavs, probs = [], [], []
    cur = [0]
    with VADWrapper(config=cfg) as vad:
        vad.set_callbacks(lambda: log.append((1, cur[0])), lambda w: (log.append((2, cur[0])), wavs.append(w)),
                          lambda b: None)
        for i in range(speech.size // 480):
            cur[0] = i
            vad.process_audio_data(speech[i * 480:(i + 1) * 480].astype(np.float32) / 32767.0)
            probs.append(vad.processor.voice_probabilities[-1])
        stats = vad.get_statistics()
    assert np.abs(np.asarray(probs, np.float32) - g["probs"]).max() <= 2e-5      # bar: 1e-4
    assert [k for k, _ in log] == list(g["event_kinds"]) and [f for _, f in log] == list(g["event_frames"])
    assert len(wavs) == 4
    assert [len(w) for w in wavs] == list(g["wav_sizes"])
    assert [hashlib.sha256(w).hexdigest() for w in wavs] == [str(s) for s in g["wav_sha256"]]
    assert stats["total_frames_processed"] == speech.size // 480


def test_wrapper_overlapping_frames_and_shared_engine():
    """1024-sample chunks advance the LSTM 3 times (hop = 1/2 frame); two
    wrappers share one engine but keep independent state."""
    from cutter_vad_amd import VADConfig, VADWrapper
    from cutter_vad_amd import weights_io
    from oracle import oracle
    from tests.signals import make_streams
    audio = make_streams(2, 8, seed=31).reshape(2, -1)
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        om = oracle.OracleModel(f.read(), "f64")
    with VADWrapper(VADConfig()) as a, VADWrapper(VADConfig()) as b:
        assert a.processor.model.engine is b.processor.model.engine
        assert a.processor.model.slot != b.processor.model.slot
        ref_state = [np.zeros(256, np.float32), np.zeros(256, np.float32)]
        for k in range(4):
            for i, w in enumerate((a, b)):
                chunk = audio[i, k * 1024:(k + 1) * 1024]
                w.process_audio_data(chunk)
                got = list(w.processor.voice_probabilities)[-3:]
                frames = oracle.split_frames(chunk, 512, 256)
                ref = [om.step(oracle.denoise(fr), ref_state[i]) for fr in frames]
                assert np.abs(np.array(got) - np.array(ref)).max() <= 2e-5
        assert a.get_statistics()["total_frames_processed"] == 12
        info = a.processor.get_model_info()
        assert info["prediction_count"] == 12 and info["state_shape"]["state"] == (2, 1, 128)


def test_stream_batch_events_match_oracle():
    from cutter_vad_amd import StreamBatch, VADConfig
    from cutter_vad_amd import weights_io
    from oracle import oracle
    from tests.signals import make_streams
    n, T = 96, 40
    cfg = VADConfig(vad_start_probability=0.5, vad_end_probability=0.35, voice_start_frame_count=3,
                    voice_end_frame_count=4)
    # alternate loud / quiet blocks so that segments start and end
    frames = make_streams(n, T, seed=55)
    frames[:, 15:25] *= 0.01
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        om = oracle.OracleModel(f.read(), "f64")
    batch = StreamBatch(cfg)
    try:
        batch.add(n)
        st = np.zeros((n, 256), np.float32)
        sms = [oracle.StateMachine(0.5, 0.35, 0.8, 0.95, 3, 4) for _ in range(n)]
        n_start = n_end = 0
        for t in range(T):
            p, ev, seg = batch.step(frames[:, t])
            pr = om.step_batch(oracle.denoise(frames[:, t]).reshape(n, 512), st, nthreads=8)
            assert np.abs(p - pr).max() <= 2e-5
            for i in range(n):
                e_ref, s_ref = sms[i].step(float(p[i]), 1)     # same float32 probability the device saw
                assert ev[i] == e_ref and seg[i] == (s_ref if e_ref & 2 else 0), (t, i)
            n_start += int((ev & 1).sum())
            n_end += int(((ev & 2) != 0).sum())
        assert n_start > 10 and n_end > 10, "the scenario must exercise both transitions"
    finally:
        batch.close()


def test_chunk_in_one_launch_equals_frame_by_frame_and_keeps_the_abort_contract(speech):
    """A chunk's F overlapping frames go through ``vad_step_multi`` in one launch.  It must be indistinguishable
    from the reference's frame-by-frame loop: same probabilities (bit-exact: same kernel, same order), same
    callbacks, and after a callback that raises at frame i the stream is where frame-by-frame processing of
    frames 0..i leaves it (vad_wrapper.py:638-647)."""
    from cutter_vad_amd import VADConfig, VADWrapper
    cfg = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6, voice_end_frame_count=12)
    x = speech[: 16000 * 12].astype(np.float32) / 32767.0
    CH = 2048                                               # 7 frames per chunk
    logs, probs = [[], []], [[], []]
    with VADWrapper(VADConfig(**cfg)) as batched, VADWrapper(VADConfig(**cfg)) as serial:
        serial.processor.process_frames = None               # forces the per-frame path of _process_audio_frames
        for k, w in enumerate((batched, serial)):
            w.set_callbacks(lambda k=k: logs[k].append("S"), lambda wav, k=k: logs[k].append(hashlib.sha256(wav).hexdigest()),
                            lambda pcm, k=k: logs[k].append(len(pcm)))
            for i in range(x.size // CH):
                w.process_audio_data(x[i * CH:(i + 1) * CH])
                probs[k].extend(list(w.processor.voice_probabilities)[-7:])
        assert probs[0] == probs[1] and len(probs[0]) == 7 * (x.size // CH)
        assert logs[0] == logs[1] and "S" in logs[0] and any(isinstance(e, str) and len(e) == 64 for e in logs[0])
        eng = batched.processor.model.engine
        assert np.array_equal(eng.get_state(batched.processor.model.slot), eng.get_state(serial.processor.model.slot))
        assert eng.save_stream(batched.processor.model.slot) == eng.save_stream(serial.processor.model.slot)

        # abort: START fires inside a chunk and its callback raises
        def boom():
            raise RuntimeError("cb")
        for w in (batched, serial):
            w.reset()
            w.set_callbacks(voice_start_callback=boom)
        from cutter_vad_amd import AudioProcessingError
        failed_at = None
        for i in range(x.size // CH):
            outcome = []
            for w in (batched, serial):
                try:
                    w.process_audio_data(x[i * CH:(i + 1) * CH])
                    outcome.append(None)
                except AudioProcessingError as e:
                    outcome.append(str(e))
            assert outcome[0] == outcome[1]
            if outcome[0] is not None:
                failed_at = i
                break
        assert failed_at is not None and "cb" in outcome[0]
        assert eng.save_stream(batched.processor.model.slot) == eng.save_stream(serial.processor.model.slot)
        assert batched.get_statistics()["total_frames_processed"] == serial.get_statistics()["total_frames_processed"]
        assert batched.is_voice_active() and serial.is_voice_active()


def test_async_wrapper_on_the_real_engine_equals_the_sync_wrapper():
    """f4: AsyncVADWrapper (src/real_time_vad/core/async_vad_wrapper.py:214-229) over the HIP engine: the audio goes through the
    executor thread, coroutine callbacks come back on the submitting loop; events, payloads and probabilities are those of the
    synchronous wrapper on the same audio (1 024-sample chunks: three overlapping frames per call, one launch each)."""
    import asyncio
    import hashlib
    import os
    from cutter_vad_amd import AsyncVADWrapper, VADConfig, VADWrapper
    pcm = np.load(os.path.join(os.path.dirname(__file__), "golden", "speech16k_i16.npz"))["pcm"]
    sp = (pcm.astype(np.float32) / np.float32(32767.0))[: 120 * 1024].reshape(120, 1024)
    cfg = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=4, voice_end_frame_count=10)
    want = []
    with VADWrapper(VADConfig(**cfg)) as w:
        w.set_callbacks(lambda: want.append("S"), lambda b: want.append(("E", hashlib.sha256(b).hexdigest())),
                        lambda b: want.append(("C", len(b))))
        for c in sp:
            w.process_audio_data(c)
        want_p = list(w.processor.voice_probabilities)

    async def run():
        got = []
        aw = AsyncVADWrapper(VADConfig(**cfg))

        async def on_start():
            got.append("S")

        async def on_end(b):
            await asyncio.sleep(0)
            got.append(("E", hashlib.sha256(b).hexdigest()))

        async def on_cont(b):
            got.append(("C", len(b)))
        aw.set_async_callbacks(on_start, on_end, on_cont)
        try:
            for c in sp:
                await aw.process_audio_data_async(c)
                await asyncio.sleep(0)
            for _ in range(20):                     # let the posted coroutines finish
                await asyncio.sleep(0.005)
            stats = await aw.get_statistics_async()
            probs = list(aw.vad_wrapper.processor.voice_probabilities)
            active = await aw.is_voice_active_async()
        finally:
            res = aw.cleanup()
            if asyncio.iscoroutine(res):
                await res
        return got, stats, probs, active

    got, stats, probs, _ = asyncio.run(run())
    assert probs == want_p and stats["total_frames_processed"] == 360
    assert [g for g in got if g == "S" or g[0] == "E"] == [x for x in want if x == "S" or x[0] == "E"]
    assert sum(1 for g in got if g != "S" and g[0] == "C") == sum(1 for x in want if x != "S" and x[0] == "C")
    assert any(g != "S" and g[0] == "E" for g in got)


def test_stream_batch_steps_mixed_rate_streams():
    """StreamBatch.step_rates: the multi-stream caller's form of configs[3] (8 / 24 / 48 kHz chunks -> GPU resample -> V5)."""
    from cutter_vad_amd import StreamBatch, VADConfig
    from cutter_vad_amd import weights_io
    from oracle import oracle
    from tests.signals import make_streams
    rates = ((8000, 256), (24000, 768), (48000, 1536))
    per, T = 40, 6
    base = make_streams(3 * per, 3 * T, seed=77).reshape(3 * per, -1)
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        om = oracle.OracleModel(f.read(), "f64")
    batch = StreamBatch(VADConfig())
    try:
        batch.add(3 * per)
        st = np.zeros((3 * per, 256), np.float32)
        for t in range(T):
            segs = [(np.ascontiguousarray(base[k * per:(k + 1) * per, t * n_in:(t + 1) * n_in]), sr) for k, (sr, n_in) in enumerate(rates)]
            p, ev, seg = batch.step_rates(segs)
            x16 = np.stack([oracle.resample(a[i], 512) for a, _ in segs for i in range(per)]).astype(np.float32)
            ref = om.step_batch(oracle.denoise(x16).reshape(3 * per, 512), st, nthreads=8)
            assert np.abs(p - ref).max() <= 5e-5 and ev.shape == (3 * per,)
    finally:
        batch.close()

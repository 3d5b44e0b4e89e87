"""GPU parity of Silero V5's 8 kHz sub-model (a9 / f3): silero_v5_step<., K8 = true> through the C ABI - native 8 kHz audio in
256-sample frames - against the f64 oracle and the interpreter goldens (tests/golden/model_v5_8k.npz, confirmed by PyTorch's
operators: tools/torch_crosscheck.py)."""

import os

import numpy as np
import pytest

from cutter_vad_amd import weights_io
from tests.signals import gate, make_streams, model_cases_8k

pytestmark = pytest.mark.gpu
TOL_P = 2e-5      # bar: 1e-4
STATE_REL = 2e-4  # final (h, c), relative to max(1, |ref|): fixed (see test_interpreter_goldens_gate_int16_and_edges)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def blob():
    with open(weights_io.packaged_blob_path(5, 8000), "rb") as f:
        return f.read()


@pytest.fixture(scope="module", params=[16, 32], ids=["tile16", "tile32"])
def engine(blob, request):
    """both kernel shapes: silero_v5_step16<., ., K8> (16-stream tiles: what the engine picks for calls of at most 4 096 streams) and
    silero_v5_step<., K8> (32-stream tiles), pinned with vad_debug_set_tile"""
    from cutter_vad_amd.engine import Engine
    e = Engine(blob, model_version=5, max_streams=2048, sample_rate=8000)
    e.set_tile(request.param)
    yield e
    e.close()


@pytest.fixture(scope="module")
def om(blob):
    from oracle import oracle
    return oracle.OracleModel(blob, "f64")


def continuation_8k(T=200):
    """200 further 256-sample frames that keep the state in play: 50-frame blocks of the harmonic "voice" and of quiet noise."""
    x = gate(make_streams(2, T // 2, seed=4040).reshape(2, T, 256))
    out = np.empty((T, 256), np.float32)
    for b in range(0, T, 50):
        out[b:b + 50] = x[(b // 50 + 1) % 2, b:b + 50]
    return out


def _frames(n, T, seed):
    return make_streams(n, (T + 1) // 2, seed=seed).reshape(n, -1, 256)[:, :T]


@pytest.mark.parametrize("n", [1, 7, 32, 33, 200, 1000])
def test_step_matches_oracle(engine, om, n):
    from oracle import oracle
    T = 10
    x = _frames(n, T, 500 + n)
    assert engine.info()["frame_samples"] == 256 and engine.info()["sample_rate"] == 8000
    slots = engine.open_streams(n)
    try:
        st = np.zeros((n, 256), np.float32)
        for t in range(T):
            got = engine.step(slots, x[:, t])
            ref = om.step_batch(oracle.denoise(x[:, t]).reshape(n, 256), st, nthreads=8)
            assert np.abs(got - ref).max() <= TOL_P, t
        dev = np.stack([engine.get_state(int(s)) for s in slots[:16]])
        assert np.abs(dev - st[:16]).max() <= 2e-4
        # T frames in one launch: same bits
        engine.reset(slots)
        one = np.stack([engine.step(slots, x[:, t]) for t in range(T)], axis=1)
        engine.reset(slots)
        multi, _ = engine.step_multi(slots, x)
        assert np.array_equal(one, multi)
    finally:
        for s in slots:
            engine.close_stream(int(s))


def test_interpreter_goldens_gate_int16_and_edges(engine, om, blob):
    from oracle import oracle
    g = np.load(os.path.join(GOLD, "model_v5_8k.npz"))
    pcm = np.load(os.path.join(GOLD, "speech16k_i16.npz"))["pcm"]
    cases = model_cases_8k(pcm)
    cont = continuation_8k()
    s = engine.open_stream()
    try:
        for name in ("speech_gate", "noise_0.3", "harmonic", "zeros", "square_fullscale", "short200_padded"):
            engine.reset([s])
            got = np.array([engine.step([s], f[None], denoise=None)[0] for f in cases[name]], np.float32)   # cases are pre-gated
            assert np.abs(got - g[f"{name}.probs"]).max() <= TOL_P, name
            # Final (h, c).  c is an unbounded accumulator (|c| reaches 140 over the 530 speech frames), so the bar is relative
            # to max(1, |ref|), and it is ONE fixed number: STATE_REL = 2e-4, the figure every other state check of the suite
            # uses (absolute 2e-4 on states of magnitude <= 8).  It does not follow any measurement: the table of what the
            # kernel, the oracle's float32 build and PyTorch's float32 operators show on these six cases is
            # profiles/r04_v5_8k_state_table.json (largest: oracle-f32 1.04e-4 on `zeros`, kernel 5.6e-5 on the square wave).
            ref_s = g[f"{name}.state"]
            dev_s = engine.get_state(s)
            rel = np.abs(dev_s - ref_s) / np.maximum(1.0, np.abs(ref_s))
            assert rel.max() <= STATE_REL, (name, float(rel.max()))
            # ... and the consequence that matters: 200 further frames from the kernel's state on the device and from the
            # float64 oracle's own state on the host give the same probabilities to the ordinary bar.
            st64 = np.zeros((1, 256), np.float32)
            for f in cases[name]:
                om.step_batch(np.ascontiguousarray(f[None]), st64, nthreads=1)
            got_c = np.array([engine.step([s], f[None], denoise=None)[0] for f in cont], np.float32)
            ref_c = np.array([om.step_batch(np.ascontiguousarray(f[None]), st64, nthreads=1)[0] for f in cont], np.float32)
            assert np.abs(got_c - ref_c).max() <= TOL_P, (name, float(np.abs(got_c - ref_c).max()))
        # the gate in the kernel == the gate of the fixture generator; no gate on the ungated speech
        engine.reset([s])
        sp = (pcm[::2].astype(np.float32) / np.float32(32767.0))[: 120 * 256].reshape(120, 256)
        got = np.array([engine.step([s], f[None], denoise=None)[0] for f in sp], np.float32)
        assert np.abs(got - g["speech_nogate_first120.probs"]).max() <= TOL_P
        engine.reset([s])
        got = np.array([engine.step([s], f[None], denoise=0.01)[0] for f in sp], np.float32)
        assert np.abs(got - g["speech_gate.probs"][:120]).max() <= TOL_P
        # int16 ingest: raw PCM, both scalings, bit-identical to float32 frames of the same quantised signal
        q = pcm[::2][: 60 * 256].reshape(60, 256)
        for scale in (32767, 32768):
            engine.reset([s])
            a = np.array([engine.step([s], (f.astype(np.float32) / np.float32(scale))[None])[0] for f in q])
            engine.reset([s])
            b = np.array([engine.step([s], f[None], i16_scale=scale)[0] for f in q])
            assert np.array_equal(a, b), scale
    finally:
        engine.close_stream(s)
    # batched golden: 7 streams advanced together
    fb = gate(make_streams(7, 5, seed=78).reshape(7, 10, 256))
    slots = engine.open_streams(7)
    try:
        got = np.stack([engine.step(slots, fb[:, t], denoise=None) for t in range(10)], axis=1)
        assert np.abs(got - g["batch7.probs"]).max() <= TOL_P
    finally:
        for k in slots:
            engine.close_stream(int(k))


def test_the_two_tile_shapes_agree_and_the_engine_picks_by_call_size(blob):
    from cutter_vad_amd.engine import Engine
    x = _frames(300, 6, 77)
    got = {}
    with Engine(blob, model_version=5, max_streams=8192, sample_rate=8000) as e:
        slots = e.open_streams(300)
        for tile in (16, 32, 0):
            e.set_tile(tile)
            e.reset(slots)
            got[tile] = np.stack([e.step(slots, x[:, t]) for t in range(6)], axis=1)
        assert np.abs(got[16] - got[32]).max() <= 2e-6           # the same algebra on the two MFMA shapes: equal to rounding
        assert np.array_equal(got[0], got[16])                  # 300 streams: the engine's own choice is the 16-stream tiles
        big = e.open_streams(8192 - 300)
        allslots = np.concatenate([slots, big])
        xb = _frames(8192, 1, 5)
        e.set_tile(0)
        e.reset(allslots)
        a = e.step(allslots, xb[:, 0])
        e.set_tile(16)
        e.reset(allslots)
        assert np.array_equal(a, e.step(allslots, xb[:, 0]))    # 8 192 streams, ONE frame: 16-stream tiles, two workgroups per CU
        xm = _frames(8192, 2, 6)
        e.set_tile(0)
        e.reset(allslots)
        am = e.step_multi(allslots, xm)[0]
        e.set_tile(32)
        e.reset(allslots)
        assert np.array_equal(am, e.step_multi(allslots, xm)[0])   # 8 192 streams, two frames per launch: 32-stream tiles


def test_wrong_frame_length_and_wrong_blob_are_refused(engine, blob):
    from cutter_vad_amd import AudioProcessingError, ModelInitializationError
    from cutter_vad_amd.engine import Engine
    s = engine.open_stream()
    try:
        with pytest.raises(AudioProcessingError, match="expected"):
            engine.step([s], np.zeros((1, 512), np.float32))
    finally:
        engine.close_stream(s)
    with pytest.raises(ModelInitializationError, match="sub-model"):
        Engine(blob, model_version=5, max_streams=8, sample_rate=16000)
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        b16 = f.read()
    with pytest.raises(ModelInitializationError, match="sub-model"):
        Engine(b16, model_version=5, max_streams=8, sample_rate=8000)
    with pytest.raises(ModelInitializationError, match="resample"):
        Engine(b16, model_version=5, max_streams=8, sample_rate=48000)


def test_full_size_tile_independence_and_state_machine(engine):
    """2 048 streams (64 tiles): duplicates in different tiles / lanes give the same bits; events follow the probabilities"""
    from oracle import oracle
    n, T = 2048, 8
    x = _frames(n // 2, T, 77)
    x = np.concatenate([x, x])
    slots = engine.open_streams(n)
    try:
        engine.set_thresholds_many(slots, (0.5, 0.35, 0.8, 0.95, 2, 3))
        sms = [oracle.StateMachine(0.5, 0.35, 0.8, 0.95, 2, 3) for _ in range(64)]
        for t in range(T):
            p, ev, seg = engine.step_events(slots, x[:, t])
            assert np.array_equal(p[: n // 2], p[n // 2:]) and np.array_equal(ev[: n // 2], ev[n // 2:])
            for i, sm in enumerate(sms):
                e_ref, s_ref = sm.step(float(p[i]), 1)
                assert ev[i] == e_ref and seg[i] == (s_ref if e_ref & 2 else 0)
    finally:
        for s in slots:
            engine.close_stream(int(s))


def test_wrapper_selects_the_8k_submodel_for_256_sample_buffers(om):
    """VADConfig(sample_rate=8000, buffer_size=256): the drop-in wrapper runs native 8 kHz audio through the sub-model; the
    reference's own combination (512-sample frames at 8 kHz) raises as onnxruntime does there."""
    from cutter_vad_amd import AudioProcessingError, SampleRate, VADConfig, VADWrapper
    from oracle import oracle
    pcm = np.load(os.path.join(GOLD, "speech16k_i16.npz"))["pcm"]
    sp = (pcm[::2].astype(np.float32) / np.float32(32767.0))[: 200 * 256].reshape(200, 256)
    ev = []
    with VADWrapper(VADConfig(sample_rate=SampleRate(8000), buffer_size=256, vad_start_probability=0.5, vad_end_probability=0.3,
                              voice_start_frame_count=3, voice_end_frame_count=8)) as w:
        w.set_callbacks(voice_start_callback=lambda: ev.append("S"), voice_end_callback=lambda b: ev.append(("E", len(b))),
                        voice_continue_callback=lambda b: None)
        st = np.zeros(256, np.float32)
        for f in sp:
            w.process_audio_data(f)
            ref = om.step(oracle.denoise(f).reshape(256), st)
            assert abs(w.processor.voice_probabilities[-1] - ref) <= TOL_P
        assert w.processor.get_model_info()["state_shape"]["state"] == (2, 1, 128)
    assert ev and ev[0] == "S" and any(isinstance(e, tuple) for e in ev)
    with VADWrapper(VADConfig(sample_rate=SampleRate(8000), buffer_size=512)) as w:
        with pytest.raises(AudioProcessingError, match="Model prediction failed"):
            w.process_audio_data(np.zeros(512, np.float32))

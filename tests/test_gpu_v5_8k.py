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
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def blob():
    with open(weights_io.packaged_blob_path(5, 8000), "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def engine(blob):
    from cutter_vad_amd.engine import Engine
    e = Engine(blob, model_version=5, max_streams=2048, sample_rate=8000)
    yield e
    e.close()


@pytest.fixture(scope="module")
def om(blob):
    from oracle import oracle
    return oracle.OracleModel(blob, "f64")


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


def test_interpreter_goldens_gate_int16_and_edges(engine, om):
    from oracle import oracle
    g = np.load(os.path.join(GOLD, "model_v5_8k.npz"))
    pcm = np.load(os.path.join(GOLD, "speech16k_i16.npz"))["pcm"]
    cases = model_cases_8k(pcm)
    s = engine.open_stream()
    try:
        for name in ("speech_gate", "noise_0.3", "harmonic", "zeros", "square_fullscale", "short200_padded"):
            engine.reset([s])
            got = np.array([engine.step([s], f[None], denoise=None)[0] for f in cases[name]], np.float32)   # cases are pre-gated
            assert np.abs(got - g[f"{name}.probs"]).max() <= TOL_P, name
            assert np.abs(engine.get_state(s) - g[f"{name}.state"]).max() <= 2e-4, name
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

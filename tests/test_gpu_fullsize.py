"""BASELINE.json's full size (8 192 streams per GPU) through the C ABI: size-independent properties.

At this size the HIP path is checked by
  * EVERY one of 8 192 distinct streams x 6 frames against the float64 oracle, probabilities and final state (the oracle runs
    0.1 - 0.2 M frames/s on the host's cores: 49 152 frames are a fraction of a second per model),
  * tile-position independence (a stream's result does not depend on which workgroup / lane serves it, bit-exact),
  * duplicate streams (same audio in two slots -> bit-identical probabilities and state),
  * T frames in one launch == T launches of one frame (bit-exact, probabilities and state),
  * int16 ingest == float32 ingest of the same quantised signal (V5: bit-exact; V4: to rounding, 2e-6),
  * the device state machine replayed by the oracle's on the device's own probabilities (exact events),
  * reset -> the run repeats bit-exactly.
Bit-exactness is the right bar for these: they compare the kernel with itself on identical arithmetic.
"""

import os

import numpy as np
import pytest

from cutter_vad_amd import weights_io
from tests.signals import make_streams

pytestmark = pytest.mark.gpu

B = 8192
TOL_P = 2e-5


@pytest.fixture(scope="module", params=[5, 4])
def setup(request):
    from cutter_vad_amd.engine import Engine
    from oracle import oracle
    v = request.param
    with open(weights_io.packaged_blob_path(v), "rb") as f:
        blob = f.read()
    eng = Engine(blob, model_version=v, max_streams=B)
    slots = eng.open_streams(B)
    yield v, eng, slots, oracle.OracleModel(blob, "f64")
    eng.close()


@pytest.fixture(scope="module")
def frames():
    T = 6
    x = make_streams(B // 2, T, seed=7)
    return np.concatenate([x, x], axis=0)          # stream i + 4096 hears what stream i hears


def test_every_stream_of_the_full_batch_matches_the_oracle(setup):
    """configs[2] (and V4's 8 192): all 8 192 DISTINCT streams x 6 frames, free-running from zero state, against the float64
    oracle - probabilities of every frame and the final (h, c) of every stream.  Same bars as the small tests."""
    from oracle import oracle
    v, eng, slots, om = setup
    T = 6
    x = make_streams(B, T, seed=11)
    eng.reset(slots)
    got = np.stack([eng.step(slots, x[:, t]) for t in range(T)], axis=1)
    assert np.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
    st = np.zeros((B, 256), np.float32)
    ref = np.empty_like(got)
    for t in range(T):
        ref[:, t] = om.step_batch(oracle.denoise(x[:, t]).reshape(B, 512), st, nthreads=os.cpu_count() or 8)
    err = np.abs(got - ref)
    worst = float(err.max())
    assert worst <= (TOL_P if v == 5 else 3e-5), (worst, np.unravel_index(err.argmax(), err.shape))
    got_state = np.stack([eng.get_state(int(s)) for s in slots])
    assert np.abs(got_state - st).max() <= 2e-4
    print(f"V{v}: {B} streams x {T} frames vs f64 oracle: max |dp| {worst:.2e}, mean {err.mean():.2e}, "
          f"max |d state| {np.abs(got_state - st).max():.2e}")


def test_duplicate_streams_are_identical(setup, frames):
    v, eng, slots, om = setup
    eng.reset(slots)
    T = frames.shape[1]
    got = np.stack([eng.step(slots, frames[:, t]) for t in range(T)], axis=1)
    # duplicates: different tiles, different lanes, same bits
    assert np.array_equal(got[: B // 2], got[B // 2:])
    for i in (5, 1000, 4095):
        assert np.array_equal(eng.get_state(int(slots[i])), eng.get_state(int(slots[i + B // 2])))


def test_tile_position_independence(setup, frames):
    v, eng, slots, _ = setup
    T = 3
    eng.reset(slots)
    if v == 5:      # V5 serves small calls on 16-stream tiles (same results to rounding: tests/test_gpu_v5_t16.py); bit-exactness
        eng.set_tile(32)    # across batch sizes is a property of ONE kernel shape, so pin it for this comparison
    full = np.stack([eng.step(slots, frames[:, t]) for t in range(T)], axis=1)
    # the same audio served alone, and as a ragged batch of 45 placed on other slots
    pick = [0, 31, 32, 4095, 4097, 8191]
    for i in pick:
        eng.reset(slots[:1])
        alone = np.array([eng.step(slots[:1], frames[i:i + 1, t])[0] for t in range(T)])
        assert np.array_equal(alone, full[i]), i
    sub = np.arange(100, 145)
    eng.reset(slots[:45])
    ragged = np.stack([eng.step(slots[:45], frames[sub, t]) for t in range(T)], axis=1)
    assert np.array_equal(ragged, full[sub])
    if v == 5:
        eng.set_tile(0)
        eng.reset(slots[:45])
        small = np.stack([eng.step(slots[:45], frames[sub, t]) for t in range(T)], axis=1)    # 45 streams: 16-stream tiles
        assert np.abs(small - full[sub]).max() <= 2e-6


def test_multi_frame_launch_equals_single_frame_launches(setup, frames):
    v, eng, slots, _ = setup
    T = frames.shape[1]
    # V5: the engine serves one-frame calls on 16-stream tiles and a multi-frame call of 8 192 streams on 32-stream tiles (equal to
    # rounding); bit-exactness between the two launch forms is a property of ONE kernel shape, so each shape is pinned in turn
    for tile in ((32, 16) if v == 5 else (0,)):
        eng.set_tile(tile)
        try:
            eng.reset(slots)
            single = np.stack([eng.step(slots, frames[:, t]) for t in range(T)], axis=1)
            s_single = eng.get_state(int(slots[777]))
            eng.reset(slots)
            multi, _ = eng.step_multi(slots, frames)
            assert np.array_equal(multi, single), tile
            assert np.array_equal(eng.get_state(int(slots[777])), s_single), tile
        finally:
            eng.set_tile(0)


def test_int16_ingest_equals_float_ingest_of_the_same_samples(setup, frames):
    """The kernel's int16 decode is a true division (bit-identical to numpy's) and the fold that follows is compiled
    without FMA contraction, so the int16 and f32 instantiations of the V5 kernel round identically."""
    v, eng, slots, _ = setup
    q = np.clip(np.round(frames[:, :2] * 32767.0), -32768, 32767).astype(np.int16)
    for scale in (32767, 32768):
        xf = (q.astype(np.float32) / np.float32(scale)).astype(np.float32)
        eng.reset(slots)
        a = np.stack([eng.step(slots, xf[:, t]) for t in range(2)], axis=1)
        eng.reset(slots)
        b = np.stack([eng.step(slots, q[:, t], i16_scale=scale) for t in range(2)], axis=1)
        assert np.array_equal(a, b) if v == 5 else np.abs(a - b).max() <= 2e-6, scale


def test_device_state_machines_follow_the_oracle_at_full_size(setup):
    from oracle import oracle
    v, eng, slots, _ = setup
    T = 36
    x = make_streams(B // 8, T, seed=21)
    x[:, 14:24] *= 0.01                               # a quiet block: segments end and start again
    x = np.tile(x, (8, 1, 1))
    eng.reset(slots)
    eng.set_thresholds_many(slots, (0.5, 0.35, 0.8, 0.95, 3, 4))          # one launch for the 8 192 slots
    sms = [oracle.StateMachine(0.5, 0.35, 0.8, 0.95, 3, 4) for _ in range(B // 8)]
    starts = ends = 0
    try:
        for t in range(T):
            p, ev, seg = eng.step_events(slots, x[:, t])
            for k in range(1, 8):                      # the 8 copies agree bit for bit
                assert np.array_equal(ev[: B // 8], ev[k * (B // 8):(k + 1) * (B // 8)])
            for i, sm in enumerate(sms):
                e_ref, s_ref = sm.step(float(p[i]), 1)
                assert ev[i] == e_ref and seg[i] == (s_ref if e_ref & 2 else 0), (t, i)
            starts += int((ev & 1).sum())
            ends += int(((ev & 2) != 0).sum())
        assert starts > 1000 and ends > 1000, (starts, ends)
    finally:
        eng.set_thresholds_many(slots, (0.7, 0.7, 0.8, 0.95, 10, 50))


def test_reset_repeats_the_run(setup, frames):
    v, eng, slots, _ = setup
    eng.reset(slots)
    a = eng.step(slots, frames[:, 0])
    b = eng.step(slots, frames[:, 1])
    eng.reset(slots)
    assert np.array_equal(eng.step(slots, frames[:, 0]), a)
    assert np.array_equal(eng.step(slots, frames[:, 1]), b)
    assert not np.array_equal(a, b)

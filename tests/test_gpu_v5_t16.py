"""The 16-stream tile kernel (csrc/silero_v5_t16.hip) through the C ABI: same network on v_mfma_f32_16x16x4_f32 tiles, chosen
for calls of at most 4 096 streams.  Checked against the f64 oracle, against the 32-stream tile kernel (same results to
rounding: the two shapes associate the sums differently), and for the bit-exact self-consistency properties."""

import numpy as np
import pytest

from cutter_vad_amd import weights_io
from tests.signals import make_streams

pytestmark = pytest.mark.gpu
TOL_P = 2e-5      # bar: 1e-4


@pytest.fixture(scope="module")
def blob():
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def eng(blob):
    from cutter_vad_amd.engine import Engine
    e = Engine(blob, model_version=5, max_streams=8192)
    yield e
    e.set_tile(0)
    e.close()


@pytest.mark.parametrize("n", [1, 7, 16, 17, 33, 200, 1024])
def test_t16_matches_oracle_and_the_32_stream_tiles(eng, blob, n):
    from oracle import oracle
    om = oracle.OracleModel(blob, "f64")
    T = 8
    x = make_streams(n, T, seed=600 + n)
    slots = eng.open_streams(n)
    try:
        eng.set_tile(16)
        st = np.zeros((n, 256), np.float32)
        got = np.empty((n, T), np.float32)
        for t in range(T):
            got[:, t] = eng.step(slots, x[:, t])
            ref = om.step_batch(oracle.denoise(x[:, t]).reshape(n, 512), st, nthreads=8)
            assert np.abs(got[:, t] - ref).max() <= TOL_P, t
        dev16 = np.stack([eng.get_state(int(s)) for s in slots[:8]])
        assert np.abs(dev16 - st[:8]).max() <= 2e-4
        # T frames in one launch; int16 ingest of the quantised signal; events
        eng.reset(slots)
        multi, ev = eng.step_multi(slots, x)
        assert np.array_equal(multi, got)
        q = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        xf = (q.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
        eng.reset(slots)
        a, _ = eng.step_multi(slots, xf)
        eng.reset(slots)
        b, _ = eng.step_multi(slots, q)
        assert np.array_equal(a, b)
        # the other tile shape: same network, sums associated differently
        eng.set_tile(32)
        eng.reset(slots)
        big, ev32 = eng.step_multi(slots, x)
        assert np.abs(big - got).max() <= 2e-6
        dev32 = np.stack([eng.get_state(int(s)) for s in slots[:8]])
        assert np.abs(dev32 - dev16).max() <= 2e-5
    finally:
        eng.set_tile(0)
        for s in slots:
            eng.close_stream(int(s))


def test_t16_tile_position_independence_state_machine_and_default_policy(eng):
    from oracle import oracle
    n, T = 4096, 6
    x = make_streams(n // 2, T, seed=9)
    x = np.concatenate([x, x])
    slots = eng.open_streams(n)
    try:
        eng.set_tile(0)                          # 4 096 streams: the default policy picks 16-stream tiles
        eng.set_thresholds_many(slots, (0.5, 0.35, 0.8, 0.95, 2, 3))
        sms = [oracle.StateMachine(0.5, 0.35, 0.8, 0.95, 2, 3) for _ in range(48)]
        dflt = []
        for t in range(T):
            p, ev, seg = eng.step_events(slots, x[:, t])
            assert np.array_equal(p[: n // 2], p[n // 2:]) and np.array_equal(ev[: n // 2], ev[n // 2:])
            for i, sm in enumerate(sms):
                e_ref, s_ref = sm.step(float(p[i]), 1)
                assert ev[i] == e_ref and seg[i] == (s_ref if e_ref & 2 else 0)
            dflt.append(p.copy())
        eng.reset(slots)
        eng.set_tile(16)
        forced = [eng.step(slots, x[:, t]) for t in range(T)]
        assert all(np.array_equal(a, b) for a, b in zip(dflt, forced))
        # scattered slots / a different position in the launch: same bits
        eng.reset(slots)
        perm = np.random.default_rng(3).permutation(n)
        again = [eng.step(slots[perm], x[perm, t]) for t in range(T)]
        assert all(np.array_equal(a[perm], b) for a, b in zip(forced, again))
    finally:
        eng.set_tile(0)
        eng.set_thresholds_many(slots, (0.7, 0.7, 0.8, 0.95, 10, 50))
        for s in slots:
            eng.close_stream(int(s))

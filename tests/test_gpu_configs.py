"""BASELINE.json configs that no other GPU test runs at their own size.

configs[1]: batch = 1 024 concurrent streams, Silero V5, 16 kHz, one MI355X.
configs[4]: the per-GPU share of the 65 536-stream job - 4 096 Silero V4 + 4 096 Silero V5 streams co-resident on one GPU,
            two engines stepping concurrently on two HIP streams (the 8-GPU part is 8 independent copies of this: no
            collective).
Checks: a sample of streams against the f64 oracle, and bit-identity with the same engine running alone (concurrency and
batch size must not change a single bit: streams are independent).
"""

import numpy as np
import pytest

from cutter_vad_amd import weights_io
from tests.signals import make_streams

pytestmark = pytest.mark.gpu
TOL = {5: 2e-5, 4: 3e-5}      # bar: 1e-4


def _blob(v):
    with open(weights_io.packaged_blob_path(v), "rb") as f:
        return f.read()


def _oracle_sample(v, frames, pick):
    from oracle import oracle
    om = oracle.OracleModel(_blob(v), "f64")
    st = np.zeros((pick.size, 256), np.float32)
    out = np.empty((pick.size, frames.shape[1]), np.float32)
    for t in range(frames.shape[1]):
        out[:, t] = om.step_batch(oracle.denoise(frames[pick, t]).reshape(pick.size, 512), st, nthreads=8)
    return out, st


def test_config1_batch_1024_v5():
    from cutter_vad_amd.engine import Engine
    B, T = 1024, 12
    frames = make_streams(B, T, seed=1234)                # BASELINE.md §4 generator, the bench's seed
    with Engine(_blob(5), model_version=5, max_streams=B) as eng:
        slots = eng.open_streams(B)
        got = np.stack([eng.step(slots, frames[:, t]) for t in range(T)], axis=1)
        assert got.shape == (B, T) and np.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
        ref, ref_s = _oracle_sample(5, frames, np.arange(B))          # all 1 024 streams
        assert np.abs(got - ref).max() <= TOL[5]
        st = np.stack([eng.get_state(int(s)) for s in slots[::16]])
        assert np.abs(st - ref_s[::16]).max() <= 2e-4
        # the same 1 024 streams as 12 frames in ONE launch, and in a pool 8 x larger with scattered slots: same bits
        eng.reset(slots)
        multi, _ = eng.step_multi(slots, frames)
        assert np.array_equal(multi, got)
    with Engine(_blob(5), model_version=5, max_streams=8192) as big:
        all_slots = big.open_streams(8192)
        scattered = np.random.default_rng(5).permutation(all_slots)[:B]
        again = np.stack([big.step(scattered, frames[:, t]) for t in range(T)], axis=1)
        assert np.array_equal(again, got)


def test_config4_share_v4_and_v5_engines_concurrent_on_two_hip_streams():
    import torch
    from cutter_vad_amd import _ffi
    from cutter_vad_amd.engine import Engine
    B, T = 4096, 6
    frames = make_streams(2 * B, T, seed=1234)
    dev = torch.device("cuda:0")
    d_frames = torch.from_numpy(frames).to(dev)                       # [2B, T, 512]
    # VAD_ENGINE_SHARED_GPU: V5 keeps to 32-stream tiles (4 096 streams = 128 CUs) so that V4's 128 workgroups run beside it
    with Engine(_blob(5), model_version=5, max_streams=B, shared_gpu=True) as e5, \
            Engine(_blob(4), model_version=4, max_streams=B, shared_gpu=True) as e4:
        e5.open_streams(B)
        e4.open_streams(B)
        s5, s4 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        p5 = torch.zeros(T, B, device=dev)
        p4 = torch.zeros(T, B, device=dev)
        ev5 = torch.zeros(T, B, dtype=torch.uint8, device=dev)
        ev4 = torch.zeros(T, B, dtype=torch.uint8, device=dev)
        x5 = [d_frames[:B, t].contiguous() for t in range(T)]
        x4 = [d_frames[B:, t].contiguous() for t in range(T)]
        torch.cuda.synchronize()
        for t in range(T):          # both engines enqueue frame t back to back: the two kernels share the GPU
            e5.step_device(B, x5[t].data_ptr(), p5[t].data_ptr(), d_events=ev5[t].data_ptr(), stream=s5.cuda_stream)
            e4.step_device(B, x4[t].data_ptr(), p4[t].data_ptr(), d_events=ev4[t].data_ptr(), stream=s4.cuda_stream)
        s5.synchronize()
        s4.synchronize()
        got5, got4 = p5.cpu().numpy().T, p4.cpu().numpy().T           # [B, T]
        st5 = e5.get_state(77)
        st4 = e4.get_state(77)
    pick = np.random.default_rng(11).choice(B, 128, replace=False)
    ref5, _ = _oracle_sample(5, frames[:B], pick)
    ref4, _ = _oracle_sample(4, frames[B:], pick)
    assert np.abs(got5[pick] - ref5).max() <= TOL[5]
    assert np.abs(got4[pick] - ref4).max() <= TOL[4]
    # alone, through the host-pointer API: bit-identical probabilities and recurrent state
    for v, got, st, x in ((5, got5, st5, frames[:B]), (4, got4, st4, frames[B:])):
        with Engine(_blob(v), model_version=v, max_streams=B, shared_gpu=True) as solo:     # same kernel shape as above
            slots = solo.open_streams(B)
            alone = np.stack([solo.step(slots, x[:, t]) for t in range(T)], axis=1)
            assert np.array_equal(alone, got), v
            assert np.array_equal(solo.get_state(int(slots[77])), st), v
    assert _ffi.VAD_OK == 0


def test_config3_mixed_input_rates_one_product_call():
    """configs[3]: batch = 4 096 streams with 8 / 24 / 48 kHz input (+ a 16 kHz pass-through segment) -> on-GPU resample -> V5,
    as ONE call of the C ABI (vad_step_rates: the auto_convert_sample_rate path the reference leaves as `pass`,
    vad_wrapper.py:621-624).  Checked against the oracle chain scipy-style Fourier resample -> gate -> model on a sample of
    streams, against the two-call form (vad_resample + vad_step) bit for bit, and through device pointers."""
    import torch
    from cutter_vad_amd.engine import Engine
    from oracle import oracle
    rates = ((8000, 256), (24000, 768), (48000, 1536), (16000, 512))
    per, T = 1024, 5
    B = per * len(rates)
    rng = np.random.default_rng(33)
    base = make_streams(B, T * 3, seed=1234).reshape(B, -1)           # enough samples for the 48 kHz chunks
    om = oracle.OracleModel(_blob(5), "f64")
    with Engine(_blob(5), model_version=5, max_streams=B) as eng, Engine(_blob(5), model_version=5, max_streams=B) as two:
        slots = eng.open_streams(B)
        slots2 = two.open_streams(B)
        pick = np.concatenate([k * per + rng.choice(per, 24, replace=False) for k in range(len(rates))])
        st = np.zeros((pick.size, 256), np.float32)
        d_x = []
        for t in range(T):
            segs = [(np.ascontiguousarray(base[k * per:(k + 1) * per, t * n_in:(t + 1) * n_in]), sr) for k, (sr, n_in) in enumerate(rates)]
            p, ev, seg = eng.step_rates(segs, slots)
            # oracle chain on the sample
            x16 = np.stack([oracle.resample(segs[i // per][0][i % per], 512) if rates[i // per][0] != 16000 else segs[i // per][0][i % per]
                            for i in pick]).astype(np.float32)
            ref = om.step_batch(oracle.denoise(x16).reshape(pick.size, 512), st, nthreads=8)
            assert np.abs(p[pick] - ref).max() <= 5e-5, t            # bar 1e-4; the resampler adds <= 1e-5 on the samples
            # the same tick as separate resample and step calls on another engine: the fused launch (16 x 16 x 4 tiles in both the
            # resampler and the model) equals it to rounding; with 32-stream tiles pinned the product call is the two-launch
            # chain itself and the bits are identical
            f16 = np.concatenate([two.resample(a, sr) if sr != 16000 else a for a, sr in segs])
            p2, ev2, _ = two.step_events(slots2, f16)
            assert np.abs(p - p2).max() <= 5e-6 and (ev == ev2).mean() > 0.999
            d_x.append(segs)
        eng.reset(slots)
        two.reset(slots2)
        eng.set_tile(32)
        two.set_tile(32)
        for t in range(T):
            p, ev, _ = eng.step_rates(d_x[t], slots)
            f16 = np.concatenate([two.resample(a, sr) if sr != 16000 else a for a, sr in d_x[t]])
            p2, ev2, _ = two.step_events(slots2, f16)
            assert np.array_equal(p, p2) and np.array_equal(ev, ev2)
        eng.set_tile(0)
        two.set_tile(0)
        fused = []
        eng.reset(slots)
        for t in range(T):
            fused.append(eng.step_rates(d_x[t], slots)[0])
        # device pointers, asynchronous: replay the 5 ticks from zero state
        eng.reset(slots)
        dev = torch.device("cuda:0")
        d_p = torch.zeros(T, B, device=dev)
        ts = torch.cuda.Stream(dev)
        keep = []
        for t in range(T):
            d_segs = []
            for a, sr in d_x[t]:
                ta = torch.from_numpy(a).to(dev)
                keep.append(ta)
                d_segs.append((ta.data_ptr(), a.shape[0], sr))
            torch.cuda.synchronize()
            eng.step_rates_device(d_segs, d_p[t].data_ptr(), stream=ts.cuda_stream)
        ts.synchronize()
        for t in range(T):                      # the asynchronous device form runs the same fused launch: same bits as the host form
            assert np.array_equal(d_p[t].cpu().numpy(), fused[t]), t
        with pytest.raises(Exception, match="must hold 768"):
            eng.step_rates([(np.zeros((2, 700), np.float32), 24000)], slots[:2])
        with pytest.raises(Exception, match="supported input rates"):
            eng.step_rates([(np.zeros((2, 441), np.float32), 44100)], slots[:2])


def test_config3_at_its_stated_size_is_one_fused_launch_with_tiles_that_straddle_the_rate_boundaries():
    """configs[3] as BASELINE.json states it: 4 096 streams, a third each at 8 / 24 / 48 kHz = 1 365 + 1 365 + 1 366.  Per-segment
    padding would need 258 sixteen-stream tiles (two more than CUs, so two launches); the tiles walk the segments end to end
    instead, two of them hold the tail of one rate and the head of the next, and the tick is ONE launch of 256 tiles.  Checked:
    the oracle chain (Fourier resample -> gate -> model) on a sample that includes every stream of the straddling tiles, and bit
    equality with the same streams stepped one segment per call (pure tiles): a stream's result does not depend on its
    neighbours' rates."""
    from cutter_vad_amd.engine import Engine
    from oracle import oracle
    rates = ((8000, 256, 1365), (24000, 768, 1365), (48000, 1536, 1366))
    B, T = 4096, 4
    rng = np.random.default_rng(34)
    base = make_streams(B, T * 3, seed=4321).reshape(B, -1)
    om = oracle.OracleModel(_blob(5), "f64")
    starts = np.cumsum([0] + [r[2] for r in rates])
    # walk order inside the launch: 48 k | 8 k | 24 k -> the straddling tiles hold (48 k tail, 8 k head) and (8 k tail, 24 k head)
    edge = np.concatenate([np.arange(starts[2] + 1366 - 8, starts[2] + 1366), np.arange(0, 12), np.arange(1365 - 12, 1365),
                           np.arange(starts[1], starts[1] + 8)])
    pick = np.unique(np.concatenate([edge, rng.choice(B, 40, replace=False)]))
    seg_of = np.searchsorted(starts, pick, side="right") - 1
    with Engine(_blob(5), model_version=5, max_streams=B) as eng, Engine(_blob(5), model_version=5, max_streams=B) as solo:
        slots, slots2 = eng.open_streams(B), solo.open_streams(B)
        st = np.zeros((pick.size, 256), np.float32)
        for t in range(T):
            segs = [(np.ascontiguousarray(base[starts[k]:starts[k + 1], t * n_in:(t + 1) * n_in]), sr) for k, (sr, n_in, _) in enumerate(rates)]
            before = eng.info()["steps"]
            p, ev, _ = eng.step_rates(segs, slots)
            assert eng.info()["steps"] == before + 1                      # one launch: 256 tiles, one per CU
            x16 = np.stack([oracle.resample(segs[k][0][i - starts[k]], 512) for i, k in zip(pick, seg_of)]).astype(np.float32)
            ref = om.step_batch(oracle.denoise(x16).reshape(pick.size, 512), st, nthreads=8)
            assert np.abs(p[pick] - ref).max() <= 5e-5, t
            alone = np.concatenate([solo.step_rates([segs[k]], slots2[starts[k]:starts[k + 1]])[0] for k in range(3)])
            assert np.array_equal(p, alone), t


def test_fused_resampler_with_every_third_sample_copied_equals_the_full_contraction(monkeypatch):
    """24 / 48 kHz chunks: the fused prologue copies the samples x[3 i'] (they sit on output instants) and contracts two thirds of
    the folded operator (pack_resample_operator_t16, "P3").  VAD_RS_DENSE=1 (diagnostic, read when an engine first packs an
    operator) keeps the full contraction: the two must agree to float32 rounding on every stream, boundary tiles included."""
    from cutter_vad_amd.engine import Engine
    rates = ((8000, 256, 700), (24000, 768, 1000), (48000, 1536, 1001), (16000, 512, 300))
    B, T = sum(r[2] for r in rates), 4
    base = make_streams(B, T * 3, seed=777).reshape(B, -1)
    starts = np.cumsum([0] + [r[2] for r in rates])
    with Engine(_blob(5), model_version=5, max_streams=B) as eng, Engine(_blob(5), model_version=5, max_streams=B) as full:
        slots, slots2 = eng.open_streams(B), full.open_streams(B)
        worst = 0.0
        for t in range(T):
            segs = [(np.ascontiguousarray(base[starts[k]:starts[k + 1], t * n_in:(t + 1) * n_in]), sr) for k, (sr, n_in, _) in enumerate(rates)]
            p, ev, _ = eng.step_rates(segs, slots)
            monkeypatch.setenv("VAD_RS_DENSE", "1")
            p2, ev2, _ = full.step_rates(segs, slots2)
            monkeypatch.delenv("VAD_RS_DENSE")
            worst = max(worst, float(np.abs(p - p2).max()))
            assert (ev == ev2).mean() > 0.999
        assert 0.0 < worst <= 5e-6, worst             # not the same code path (> 0), the same answer


def test_one_tile_with_four_parts_equals_the_segments_stepped_alone():
    """Ten streams at four rates are ONE 16-stream tile with four parts (48 k | 8 k | 24 k | 16 k in the launch's walk order): each part
    resamples with its own operator (two of them with every third sample copied), columns of other parts contract zeros and keep
    their frames.  Bit-equal to each segment stepped by a call of its own, and within 5e-5 of the oracle chain."""
    from cutter_vad_amd.engine import Engine
    from oracle import oracle
    rates = ((48000, 1536, 3), (8000, 256, 2), (24000, 768, 4), (16000, 512, 1))
    B, T = sum(r[2] for r in rates), 6
    base = make_streams(B, T * 3, seed=99).reshape(B, -1)
    starts = np.cumsum([0] + [r[2] for r in rates])
    om = oracle.OracleModel(_blob(5), "f64")
    st = np.zeros((B, 256), np.float32)
    with Engine(_blob(5), model_version=5, max_streams=64) as eng, Engine(_blob(5), model_version=5, max_streams=64) as solo:
        slots, slots2 = eng.open_streams(B), solo.open_streams(B)
        for t in range(T):
            segs = [(np.ascontiguousarray(base[starts[k]:starts[k + 1], t * n_in:(t + 1) * n_in]), sr) for k, (sr, n_in, _) in enumerate(rates)]
            before = eng.info()["steps"]
            p, ev, _ = eng.step_rates(segs, slots)
            assert eng.info()["steps"] == before + 1
            alone = np.concatenate([solo.step_rates([segs[k]], slots2[starts[k]:starts[k + 1]])[0] for k in range(len(rates))])
            assert np.array_equal(p, alone), t
            x16 = np.stack([oracle.resample(segs[k][0][i], 512) if rates[k][0] != 16000 else segs[k][0][i]
                            for k in range(len(rates)) for i in range(rates[k][2])]).astype(np.float32)
            ref = om.step_batch(oracle.denoise(x16).reshape(B, 512), st, nthreads=4)
            assert np.abs(p - ref).max() <= 5e-5, t

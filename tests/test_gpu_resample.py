"""GPU parity of the resampler kernel (a11) and of config 4: mixed 8/24/48 kHz -> resample -> V5."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RATES = ((8000, 256), (24000, 768), (48000, 1536))
TOL = 1e-5   # SURVEY §8 d, config 4: resample stage <= 1e-5 abs vs scipy.signal.resample


@pytest.fixture(scope="module")
def engine():
    from cutter_vad_amd import weights_io
    from cutter_vad_amd.engine import Engine
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        e = Engine(f.read(), max_streams=2048)
    yield e
    e.close()


def test_kernel_matches_reference_fixture_and_scipy(engine):
    import scipy.signal
    g = np.load(os.path.join(GOLD, "utils.npz"))
    for sr, n_in in RATES:
        x = (0.5 * np.random.default_rng(sr).standard_normal((5, n_in))).astype(np.float32)
        got = engine.resample(x, sr)
        assert got.shape == (5, 512) and got.dtype == np.float32
        assert np.abs(got - g[f"resample_{sr}"]).max() <= TOL          # the reference's AudioUtils.resample_audio
        for n in (1, 31, 32, 33, 200):                                  # ragged tiles
            xs = (0.3 * np.random.default_rng(n).standard_normal((n, n_in))).astype(np.float32)
            ref = np.stack([scipy.signal.resample(r, 512).astype(np.float32) for r in xs])
            assert np.abs(engine.resample(xs, sr) - ref).max() <= TOL


def test_oracle_agrees_and_linearity(engine):
    from oracle import oracle
    rng = np.random.default_rng(3)
    for sr, n_in in RATES:
        a = rng.standard_normal((40, n_in)).astype(np.float32)
        b = rng.standard_normal((40, n_in)).astype(np.float32)
        ya, yb, yab = engine.resample(a, sr), engine.resample(b, sr), engine.resample(a + 2 * b, sr)
        assert np.abs(yab - (ya + 2 * yb)).max() <= 2e-5                # a linear operator
        ref = np.stack([oracle.resample(r, 512) for r in a[:6]])
        assert np.abs(ya[:6] - ref).max() <= TOL
        const = engine.resample(np.full((3, n_in), 0.25, np.float32), sr)
        assert np.abs(const - 0.25).max() <= TOL                        # DC is preserved (fp32 operator rows)


def test_both_launch_shapes_agree(engine):
    """Calls of up to 256 chunk tiles split the 512 outputs over two workgroups per tile, larger calls use one: the two
    shapes must give bit-identical rows (same operator blocks, same order of accumulation)."""
    import scipy.signal
    for sr, n_in in ((8000, 256), (48000, 1536)):
        x = (0.4 * np.random.default_rng(sr + 1).standard_normal((8300, n_in))).astype(np.float32)   # 260 tiles
        big = engine.resample(x, sr)
        small = engine.resample(x[:100], sr)                                                         # 4 tiles, split
        assert np.array_equal(big[:100], small)
        pick = [0, 31, 32, 4127, 8299]
        ref = np.stack([scipy.signal.resample(x[i], 512).astype(np.float32) for i in pick])
        assert np.abs(big[pick] - ref).max() <= TOL


def test_mixed_rates_in_one_launch(engine):
    """vad_resample_multi_device: the 8 / 24 / 48 kHz chunks of a tick in ONE launch == three single-rate calls, bit for bit."""
    import torch
    rng = np.random.default_rng(5)
    xs = [(0.3 * rng.standard_normal((n, n_in))).astype(np.float32) for n, (_, n_in) in zip((70, 33, 201), RATES)]
    ref = [engine.resample(x, sr) for x, (sr, _) in zip(xs, RATES)]
    d_in = [torch.from_numpy(x).cuda() for x in xs]
    d_out = [torch.full((x.shape[0], 512), float("nan"), device="cuda") for x in xs]
    segs = [(a.data_ptr(), a.shape[0], a.shape[1], sr, o.data_ptr()) for a, o, (sr, _) in zip(d_in, d_out, RATES)]
    engine.resample_multi_device(segs)
    engine.synchronize()
    for o, r in zip(d_out, ref):
        assert np.array_equal(o.cpu().numpy(), r)
    from cutter_vad_amd import AudioProcessingError
    with pytest.raises(AudioProcessingError):
        engine.resample_multi_device([(d_in[0].data_ptr(), 70, 256, 44100, d_out[0].data_ptr())])
    with pytest.raises(AudioProcessingError):
        engine.resample_multi_device(segs + segs)                      # more than 4 segments


def test_audio_utils_resample_goes_through_the_engine(engine):
    from cutter_vad_amd import AudioUtils
    import scipy.signal
    x = (0.5 * np.random.default_rng(11).standard_normal(1536)).astype(np.float32)
    y = AudioUtils.resample_audio(x, 48000, 16000)
    assert y.shape == (512,) and y.dtype == np.float32
    assert np.abs(y - scipy.signal.resample(x, 512).astype(np.float32)).max() <= TOL
    # more than one chunk: the WHOLE array through one transform, as the reference does it - not chunk by chunk
    xx = np.concatenate([x, x[::-1], x])
    y3 = AudioUtils.resample_audio(xx, 48000, 16000)
    assert y3.shape == (1536,) and np.abs(y3 - scipy.signal.resample(xx, 1536).astype(np.float32)).max() <= TOL
    assert np.abs(y3[:512] - y).max() > 1e-3          # and that is a different signal from three per-chunk resamples
    # the one-chunk MFMA kernel and the generic kernel are the same function of the input
    z = engine.resample_generic(x.reshape(1, -1), 512)[0]
    assert np.abs(z - y).max() <= 2e-6


def test_generic_resampler_matches_the_reference_function_on_any_shape(engine):
    """AudioUtils.resample_audio on whole arrays of arbitrary length / rates / dtype: outputs of the reference's own function
    (tests/golden/resample_generic.npz; case 1 is /root/reference/tests/test_audio_utils.py:67-82) and live scipy."""
    import warnings
    import scipy.signal
    from cutter_vad_amd import AudioUtils
    from tests.signals import resample_generic_inputs
    g = np.load(os.path.join(GOLD, "resample_generic.npz"))
    for name, x, r0, r1 in resample_generic_inputs():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = AudioUtils.resample_audio(x, r0, r1)
            live = scipy.signal.resample(np.asarray(x), int(len(x) * (r1 / r0))).astype(np.float32)
        want = g[name]
        scale = max(1.0, float(np.abs(want).max()))
        assert got.dtype == np.float32 and got.shape == want.shape, name
        assert np.abs(got - want).max() <= TOL * scale and np.abs(got - live).max() <= TOL * scale, name
    # the reference's own assertions on its own case
    t = np.linspace(0, 0.1, 100, False)
    r = AudioUtils.resample_audio(np.sin(2 * np.pi * 50 * t).astype(np.float32), 1000, 500)
    assert abs(len(r) - 50) <= 1 and r.dtype == np.float32


def test_generic_resampler_long_arrays_rows_and_launch_shapes(engine):
    import scipy.signal
    from cutter_vad_amd import AudioProcessingError, AudioUtils
    rng = np.random.default_rng(77)
    # one second of audio: 48 k -> 16 k (t = (3 m - n) / 48000) and 44.1 k -> 16 k (441 / 160: a 7 M-long period)
    for n_in, r0 in ((48000, 48000), (44100, 44100)):
        x = (0.3 * rng.standard_normal(n_in)).astype(np.float32)
        y = AudioUtils.resample_audio(x, r0, 16000)
        assert y.shape == (16000,) and np.abs(y - scipy.signal.resample(x, 16000).astype(np.float32)).max() <= TOL
    # coprime lengths: every t is distinct, the closest to 0 are ~1e-9
    x = (0.3 * rng.standard_normal(30011)).astype(np.float32)
    y = engine.resample_generic(x.reshape(1, -1), 10007)[0]
    assert np.abs(y - scipy.signal.resample(x, 10007).astype(np.float32)).max() <= TOL
    # rows: many arrays of one shape in one call == one call each, bit for bit (fixed summation order), float64 input too
    xs = (0.3 * rng.standard_normal((37, 1411))).astype(np.float32)
    for mode in (1, 2):                                   # within a kernel (pinned: the size rule looks at rows x n_in x n_out)
        engine.set_resample_path(mode)
        ys = engine.resample_generic(xs, 512)
        assert ys.shape == (37, 512)
        for i in (0, 17, 36):
            assert np.array_equal(ys[i], engine.resample_generic(xs[i:i + 1], 512)[0]), (mode, i)
    engine.set_resample_path(0)
    assert np.abs(ys - scipy.signal.resample(xs, 512, axis=1).astype(np.float32)).max() <= TOL
    xd = rng.standard_normal((3, 2000)) * 1e4
    yd = engine.resample_generic(xd, 3000)
    assert np.abs(yd - scipy.signal.resample(xd, 3000, axis=1).astype(np.float32)).max() <= 1e-6 * 1e4
    # linearity and constants (DC gain 1)
    a, b = xs[:4], xs[4:8]
    assert np.abs(engine.resample_generic(a + 2 * b, 700) - (engine.resample_generic(a, 700) + 2 * engine.resample_generic(b, 700))).max() <= 2e-6
    assert np.abs(engine.resample_generic(np.full((1, 999), 0.25, np.float32), 1234) - 0.25).max() <= 1e-6
    # NaN / Inf in one column reach every output of that column and no other (what the transform does)
    st = (0.3 * rng.standard_normal((300, 2))).astype(np.float32)
    st[5, 1] = np.inf
    ynan = AudioUtils.resample_audio(st, 48000, 16000)
    assert np.isnan(ynan[:, 1]).all() and np.abs(ynan[:, 0] - scipy.signal.resample(st[:, 0], 100)).max() <= TOL
    # beyond both kernels' limits (longer than the FFT path takes AND too many entries for the direct kernel): refused, never cut
    with pytest.raises(AudioProcessingError, match="2\\^42 operator entries"):
        engine.resample_generic(np.zeros((1, (1 << 25) + 1), np.float32), 1 << 24)
    assert np.array_equal(engine.resample_generic(np.zeros((0, 10), np.float32), 5), np.zeros((0, 5), np.float32))


def test_long_arrays_take_the_fft_path_and_both_paths_are_the_same_function(engine):
    """vad_resample_generic has two kernels behind it: every operator entry evaluated (O(n_in n_out)) for small calls, two chirp-z
    transforms on power-of-two float64 FFTs (O(n log n)) from 2^25 entries up.  Pinned to either (vad_debug_resample_path), both
    must be scipy's function - on the reference-generated fixture shapes, awkward lengths, rows, float64 input - and agree with
    each other; a ten-second and a one-minute array go through the size rule."""
    import warnings
    import scipy.signal
    from cutter_vad_amd import AudioUtils
    from tests.signals import resample_generic_inputs
    g = np.load(os.path.join(GOLD, "resample_generic.npz"))
    try:
        for mode in (2, 1):                                   # 2 = FFT path, 1 = direct kernel
            engine.set_resample_path(mode)
            from cutter_vad_amd.pool import default_pool
            default_pool().any_engine().set_resample_path(mode)            # the engine AudioUtils goes through
            for name, x, r0, r1 in resample_generic_inputs():
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    got = AudioUtils.resample_audio(x, r0, r1)
                want = g[name]
                assert got.shape == want.shape and np.abs(got - want).max() <= TOL * max(1.0, float(np.abs(want).max())), (mode, name)
        rng = np.random.default_rng(5)
        for n_in, n_out in ((1, 1), (2, 1), (1, 5), (3, 3), (8, 8), (255, 256), (256, 255), (4097, 333), (333, 4097), (30011, 10007)):
            x = (0.3 * rng.standard_normal((3, n_in))).astype(np.float32)
            ref = scipy.signal.resample(x, n_out, axis=1).astype(np.float32)
            engine.set_resample_path(2)
            a = engine.resample_generic(x, n_out)
            engine.set_resample_path(1)
            b = engine.resample_generic(x, n_out)
            assert np.abs(a - ref).max() <= 2e-6 and np.abs(b - ref).max() <= 2e-6 and np.abs(a - b).max() <= 5e-7, (n_in, n_out)
        xd = rng.standard_normal((2, 5000)) * 1e4
        engine.set_resample_path(2)
        assert np.abs(engine.resample_generic(xd, 7777) - scipy.signal.resample(xd, 7777, axis=1).astype(np.float32)).max() <= 1e-6 * 1e4
    finally:
        engine.set_resample_path(0)
        default_pool().any_engine().set_resample_path(0)
    # the size rule: ten seconds (7.7e10 entries) and a minute of 48 kHz audio, and an array longer than the direct kernel's cap
    for seconds in (10, 60):
        x = (0.3 * rng.standard_normal(48000 * seconds)).astype(np.float32)
        y = AudioUtils.resample_audio(x, 48000, 16000)
        assert y.shape == (16000 * seconds,) and np.abs(y - scipy.signal.resample(x, 16000 * seconds).astype(np.float32)).max() <= TOL
    x = (0.3 * rng.standard_normal(44100 * 120)).astype(np.float32)          # two minutes at 44.1 kHz: 1e13 entries
    y = AudioUtils.resample_audio(x, 44100, 16000)
    assert y.shape == (16000 * 120,) and np.abs(y - scipy.signal.resample(x, 16000 * 120).astype(np.float32)).max() <= TOL


def test_generic_resampler_on_random_shapes_both_kernels(engine):
    """200 random (rows, n_in, n_out, dtype) - lengths 1 .. 20 000, up- and down-sampling, equal lengths, powers of two, primes -
    through each of the two kernels: scipy's answer to float32 rounding every time."""
    import scipy.signal
    rng = np.random.default_rng(20260)
    special = [1, 2, 3, 4, 5, 7, 8, 16, 17, 64, 127, 128, 129, 255, 256, 257, 511, 512, 513, 997, 1000, 1024, 4096, 4099, 8191, 8192, 16384]
    worst = 0.0
    try:
        for k in range(200):
            n_in = int(rng.choice(special)) if rng.random() < 0.35 else int(rng.integers(1, 20001))
            n_out = n_in if rng.random() < 0.05 else (int(rng.choice(special)) if rng.random() < 0.35 else int(rng.integers(1, 20001)))
            rows = int(rng.integers(1, 5))
            wide = rng.random() < 0.3
            x = rng.standard_normal((rows, n_in)) * (10.0 if wide else 0.3)
            x = x if wide else x.astype(np.float32)
            ref = scipy.signal.resample(x, n_out, axis=1).astype(np.float32)
            scale = max(1.0, float(np.abs(ref).max()))
            for mode in (1, 2):
                engine.set_resample_path(mode)
                got = engine.resample_generic(x, n_out)
                err = float(np.abs(got - ref).max()) / scale
                worst = max(worst, err)
                assert got.shape == ref.shape and err <= 3e-6, (k, mode, rows, n_in, n_out, wide, err)
    finally:
        engine.set_resample_path(0)
    assert worst > 0.0


def test_generic_resampler_on_device_pointers(engine):
    """vad_resample_generic_device: device buffers in and out (a GPU decode pipeline's audio never visits the host), both kernels,
    float32 and float64 input; equal to the host-pointer form bit for bit."""
    import torch
    rng = np.random.default_rng(8)
    try:
        for mode, (rows, n_in, n_out) in ((1, (3, 1411, 512)), (2, (3, 1411, 512)), (2, (1, 96000, 32000))):
            engine.set_resample_path(mode)
            for dt in (np.float32, np.float64):
                x = (0.3 * rng.standard_normal((rows, n_in))).astype(dt)
                d_x = torch.from_numpy(x).cuda()
                d_y = torch.zeros(rows, n_out, device="cuda")
                torch.cuda.synchronize()
                engine.resample_generic_device(d_x.data_ptr(), rows, n_in, n_out, d_y.data_ptr(), f64=dt is np.float64)
                assert np.array_equal(d_y.cpu().numpy(), engine.resample_generic(x, n_out)), (mode, dt)
        with pytest.raises(Exception, match="null buffer"):
            engine.resample_generic_device(0, 1, 10, 5, 0)
    finally:
        engine.set_resample_path(0)


def test_config4_mixed_rates_resample_then_v5(engine):
    """4096-stream config scaled down: thirds at 8/24/48 kHz, on-GPU resample, V5; oracle on the same chain."""
    from cutter_vad_amd import weights_io
    from oracle import oracle
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        om = oracle.OracleModel(f.read(), "f64")
    per, T = 40, 5
    slots = engine.open_streams(3 * per)
    try:
        st = np.zeros((3 * per, 256), np.float32)
        rng = np.random.default_rng(8)
        for t in range(T):
            frames16, ref16 = [], []
            for k, (sr, n_in) in enumerate(RATES):
                tt = (np.arange(n_in) + t * n_in) / sr
                x = (0.3 * np.sin(2 * np.pi * (150 + 10 * np.arange(per))[:, None] * tt[None, :])
                     + 0.03 * rng.standard_normal((per, n_in))).astype(np.float32)
                frames16.append(engine.resample(x, sr))
                ref16.append(np.stack([oracle.resample(r, 512) for r in x]))
            f16 = np.concatenate(frames16)
            r16 = np.concatenate(ref16)
            assert np.abs(f16 - r16).max() <= TOL
            p = engine.step(slots, f16)
            pr = om.step_batch(oracle.denoise(r16).reshape(-1, 512), st, nthreads=8)
            assert np.abs(p - pr).max() <= 1e-4                       # the north-star bar for the full chain
    finally:
        for s in slots:
            engine.close_stream(s)

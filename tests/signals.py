"""Seeded synthetic audio used by tests and bench (BASELINE.md §4 generator)."""

import numpy as np


def make_streams(n: int, T: int, seed: int = 1234, first_stream: int = 0) -> np.ndarray:
    """[n, T, 512] float32.  Stream s is seeded default_rng(seed + s); even s: N(0, 0.02^2) noise,
    odd s: 0.3*(0.4 sin150 + 0.3 sin300 + 0.2 sin600 Hz) + N(0, 0.03^2); clipped to [-1, 1]."""
    out = np.empty((n, T * 512), np.float32)
    t = np.arange(T * 512, dtype=np.float64) / 16000.0
    for i in range(n):
        s = first_stream + i
        rng = np.random.default_rng(seed + s)
        if s % 2 == 0:
            x = 0.02 * rng.standard_normal(t.size)
        else:
            x = 0.3 * (0.4 * np.sin(2 * np.pi * 150 * t) + 0.3 * np.sin(2 * np.pi * 300 * t)
                       + 0.2 * np.sin(2 * np.pi * 600 * t)) + 0.03 * rng.standard_normal(t.size)
        out[i] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out.reshape(n, T, 512)


def gate(x, thr=0.01):
    return np.where(np.abs(x) > thr, x, 0.0).astype(np.float32)


def model_cases(speech_i16: np.ndarray):
    speech = (speech_i16.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    n = speech.size // 512
    sp = speech[: n * 512].reshape(n, 512)
    rng = np.random.default_rng(4242)
    cases = {}
    cases["speech_gate"] = dict(frames=gate(sp), regen="speech16k int16 / 32767, 512-sample frames, gate 0.01")
    cases["speech_nogate_first120"] = dict(frames=sp[:120], regen="speech16k int16 / 32767, first 120 frames, no gate")
    for sigma in (0.005, 0.02, 0.3):
        x = (sigma * np.random.default_rng(int(sigma * 1e4)).standard_normal((40, 512))).astype(np.float32)
        cases[f"noise_{sigma}"] = dict(frames=gate(x), regen=f"default_rng({int(sigma*1e4)}).standard_normal((40,512))*{sigma}, gate 0.01")
    cases["harmonic"] = dict(frames=gate(make_streams(2, 40, seed=1234)[1]), regen="tests.signals.make_streams(2,40,seed=1234)[1], gate 0.01")
    cases["zeros"] = dict(frames=np.zeros((8, 512), np.float32), regen="zeros((8,512))")
    sq = np.where(np.arange(512 * 8) % 64 < 32, 1.0, -1.0).astype(np.float32).reshape(8, 512)
    cases["square_fullscale"] = dict(frames=sq, regen="where(arange(4096)%64<32,1,-1).reshape(8,512)")
    short = (0.2 * rng.standard_normal((8, 400))).astype(np.float32)
    cases["short400_padded"] = dict(frames=np.pad(short, ((0, 0), (0, 112))), regen="default_rng(4242): 0.2*standard_normal((8,400)) right-zero-padded to 512")
    long_ = (0.2 * rng.standard_normal((8, 600))).astype(np.float32)
    cases["long600_truncated"] = dict(frames=long_[:, :512].copy(), regen="default_rng(4242) (continued): 0.2*standard_normal((8,600))[:, :512]")
    return cases


def model_cases_8k(speech_i16):
    """Inputs for Silero V5's 8 kHz sub-model: native 8 kHz audio in 256-sample frames.  The speech is the committed 16 kHz
    fixture, every second sample (tests/golden/speech16k_i16.npz -> 8 kHz); the rest are seeded."""
    sp = (speech_i16[::2].astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    n = sp.size // 256
    sp = sp[: n * 256].reshape(n, 256)
    c = {"speech_gate": gate(sp), "speech_nogate_first120": sp[:120]}
    for sigma in (0.02, 0.3):
        c[f"noise_{sigma}"] = gate((sigma * np.random.default_rng(int(sigma * 1e4) + 8).standard_normal((40, 256))).astype(np.float32))
    t = np.arange(40 * 256) / 8000.0
    h = 0.3 * (0.4 * np.sin(2 * np.pi * 150 * t) + 0.3 * np.sin(2 * np.pi * 300 * t) + 0.2 * np.sin(2 * np.pi * 600 * t)) \
        + 0.03 * np.random.default_rng(88).standard_normal(t.size)
    c["harmonic"] = gate(np.clip(h, -1, 1).astype(np.float32).reshape(40, 256))
    c["zeros"] = np.zeros((8, 256), np.float32)
    c["square_fullscale"] = np.where(np.arange(256 * 8) % 64 < 32, 1.0, -1.0).astype(np.float32).reshape(8, 256)
    short = (0.2 * np.random.default_rng(4243).standard_normal((8, 200))).astype(np.float32)
    c["short200_padded"] = np.pad(short, ((0, 0), (0, 56)))
    return c


def resample_generic_inputs():
    """(name, array, original_rate, target_rate): whole arrays for ``AudioUtils.resample_audio`` outside the three streaming
    chunk shapes.  The first case is the reference's own test (/root/reference/tests/test_audio_utils.py:67-82)."""
    def rn(seed, shape, dtype=np.float32, scale=0.5):
        return (scale * np.random.default_rng(seed).standard_normal(shape)).astype(dtype)
    t = np.linspace(0, 0.1, 100, False)
    cases = [
        ("ref_test_sine_1000_500", np.sin(2 * np.pi * 50 * t).astype(np.float32), 1000, 500),
        ("odd_1001_44100_16000", rn(1, 1001), 44100, 16000),
        ("odd_997_16000_44100", rn(2, 997), 16000, 44100),
        ("tenth_second_44100_16000", rn(3, 4410), 44100, 16000),
        ("three_chunks_48000_16000", rn(4, 4608), 48000, 16000),
        ("two_chunks_8000_16000", rn(5, 512), 8000, 16000),
        ("stereo_300x2_48000_16000", rn(6, (300, 2)), 48000, 16000),
        ("int16_480_24000_16000", (rn(7, 480, np.float64) * 20000).astype(np.int16), 24000, 16000),
        ("float64_777_22050_16000", rn(8, 777, np.float64), 22050, 16000),
        ("up_odd_255_8000_16000", rn(9, 255), 8000, 16000),
        ("up_even_256_8000_24000", rn(10, 256), 8000, 24000),
        ("near_equal_1234_16000_15990", rn(11, 1234), 16000, 15990),
        ("equal_len_even_10_16000_17599", rn(12, 10), 16000, 17599),
        ("tiny_3_48000_16000", rn(13, 3), 48000, 16000),
        ("tiny_2_16000_48000", rn(14, 2), 16000, 48000),
        ("single_1_48000_16000", rn(15, 1), 48000, 16000),
        ("down_even_out_1000_16000_8000", rn(16, 1000), 16000, 8000),
        ("prime_4093_48000_44100", rn(17, 4093), 48000, 44100),
        ("list_input_64_32000_16000", [float(v) for v in rn(18, 64, np.float64)], 32000, 16000),
    ]
    return cases


def one_ulp_conditioning(step_f64, frames, trials=256, seed=0):
    """How far the float64 ORACLE's own answer moves when every input sample is perturbed by at most one float32 ulp (relative
    2^-23): max over `trials` seeded perturbations and over the frames of |p(x (1 + d)) - p(x)|, d uniform in +-2^-23.
    `step_f64(frames[n, T, L]) -> probs[n, T]` runs n independent streams from zero state.  This is a property of the function
    and the input - its conditioning at float32 input precision - not of any float32 evaluation: an evaluation that is backward
    stable to one ulp of the input cannot be expected closer to float64 than this, and the tests that feed ill-conditioned inputs
    (exact spectral nulls under Silero V4's log(1 + |X| 2^20)) hold the kernels to exactly this bound."""
    x = np.asarray(frames, np.float32)
    base = step_f64(x[None])[0]
    d = np.random.default_rng(seed).uniform(-2.0 ** -23, 2.0 ** -23, (trials,) + x.shape)
    pert = (x[None].astype(np.float64) * (1.0 + d)).astype(np.float32)
    return float(np.abs(step_f64(pert) - base[None]).max())

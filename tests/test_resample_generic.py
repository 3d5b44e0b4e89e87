"""AudioUtils.resample_audio for arbitrary lengths and rates, without a GPU: the oracle's restatement and the HOST half of the
generic kernel (float64 tables + the entry arithmetic the kernel repeats) against outputs of the reference's own function
(tests/golden/resample_generic.npz, written by tools/make_goldens.py from /root/reference's AudioUtils) and live scipy."""

import ctypes as C
import os
import warnings

import numpy as np
import pytest

from cutter_vad_amd import AudioProcessingError, AudioUtils, _ffi
from oracle import oracle
from tests.signals import resample_generic_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def operator(n_in, n_out):
    R = np.empty((n_out, n_in), np.float64)
    assert _ffi.lib().vad_debug_resample_generic_entries(n_in, n_out, 0, n_out, R.ctypes.data_as(C.POINTER(C.c_double)), R.size) == 0
    return R


def test_oracle_and_host_tables_match_the_reference_outputs():
    g = np.load(os.path.join(GOLD, "resample_generic.npz"))
    cases = resample_generic_inputs()
    assert sorted(g.files) == sorted(c[0] for c in cases)
    for name, x, r0, r1 in cases:
        want = g[name]
        x = np.asarray(x)
        n_out = int(len(x) * (r1 / r0))
        if n_out == 0:
            assert want.shape == (1,) and want[0] == 0          # scipy's irfft(n=0) hands back one sample, times 0
            continue
        cols = x.reshape(len(x), -1).astype(np.float64)
        got = (operator(len(x), n_out) @ cols).reshape((n_out,) + x.shape[1:])
        scale = max(1.0, float(np.abs(want).max()))
        assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6 * scale, name
        if x.ndim == 1 and x.dtype == np.float32:
            assert np.abs(oracle.resample(x, n_out) - want).max() <= 2e-6, name


@pytest.mark.parametrize("n_in,n_out", [(1, 1), (1, 7), (7, 1), (2, 1), (1, 2), (2, 2), (8, 8), (9, 9), (6, 4), (4, 6), (100, 36),
                                        (255, 256), (256, 255), (2205, 800), (97, 1013), (1013, 97), (1234, 1233), (3000, 4500)])
def test_operator_entries_are_scipys_fourier_method(n_in, n_out):
    import scipy.signal
    x = np.random.default_rng(n_in * 7919 + n_out).standard_normal(n_in)
    assert np.abs(operator(n_in, n_out) @ x - scipy.signal.resample(x, n_out)).max() <= 1e-12 * max(n_in, n_out)


def test_small_angle_rule_on_a_long_coprime_pair():
    """lengths with gcd 1: t = j / (n_in n_out) comes as close to 0 as 1e-10 without being 0 - the entries next to the peak are
    where sin(a_m - b_n) as a difference of products would have lost digits"""
    n_in, n_out = 99991, 100003
    lib = _ffi.lib()
    x = np.random.default_rng(5).standard_normal(n_in)
    rows = np.empty((3, n_in), np.float64)
    import scipy.signal
    ref = scipy.signal.resample(x, n_out)
    for m0 in (0, 50000, n_out - 3):
        assert lib.vad_debug_resample_generic_entries(n_in, n_out, m0, m0 + 3, rows.ctypes.data_as(C.POINTER(C.c_double)), rows.size) == 0
        assert np.abs(rows @ x - ref[m0:m0 + 3]).max() <= 1e-9


def test_mirror_semantics_that_need_no_engine():
    x = np.arange(4, dtype=np.float32)
    assert AudioUtils.resample_audio(x, 16000, 16000) is x                       # audio.py:39-40
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = AudioUtils.resample_audio(np.array([3.0], np.float32), 48000, 16000)   # int(1 / 3) = 0 output samples
    assert one.shape == (1,) and one.dtype == np.float32 and one[0] == 0
    for args, text in (((np.zeros(0, np.float32), 48000, 16000), "invalid number of data points (0) specified"),
                       ((np.float32(3.0), 8000, 16000), "has no len()"),
                       ((np.zeros(100, np.float32), 0, 8000), "division by zero"),
                       ((np.zeros(100, np.float32), 8000, -8000), "negative dimensions are not allowed")):
        with pytest.raises(AudioProcessingError) as ei:
            AudioUtils.resample_audio(*args)
        assert str(ei.value).startswith(f"[AUDIO_PROCESSING_ERROR] Failed to resample audio from {args[1]}Hz to {args[2]}Hz: ") \
            and text in str(ei.value)


def test_debug_entry_rejects_bad_ranges():
    buf = np.empty(10, np.float64)
    p = buf.ctypes.data_as(C.POINTER(C.c_double))
    lib = _ffi.lib()
    assert lib.vad_debug_resample_generic_entries(10, 5, 0, 5, p, 10) == _ffi.VAD_ERR_INVALID_ARG       # buffer too small
    assert lib.vad_debug_resample_generic_entries(0, 5, 0, 1, p, 10) == _ffi.VAD_ERR_INVALID_ARG
    assert lib.vad_debug_resample_generic_entries(10, 5, 3, 2, p, 10) == _ffi.VAD_ERR_INVALID_ARG

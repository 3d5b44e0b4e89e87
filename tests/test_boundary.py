"""The C-ABI boundary, without a GPU: the library loads, exports every symbol the header
declares, the host-side packer produces the layout the kernel consumes, and nothing in the
product routes through the oracle or any CPU fallback."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from cutter_vad_amd import _ffi, weights_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    with open(os.path.join(ROOT, "include", "vad_engine.h")) as f:
        src = f.read()
    return re.findall(r"^VAD_API\s+[\w\s\*]+?\b(vad_\w+)\s*\(", src, flags=re.M)


def test_library_exports_every_declared_symbol():
    names = _header_symbols()
    assert len(names) >= 20
    lib = _ffi.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vad_engine.h but not exported"
    assert sorted(names) == sorted(_ffi.SIGNATURES), "ctypes table and header disagree"


def test_struct_sizes_match_header():
    assert C.sizeof(_ffi.EngineDesc) == 40
    assert C.sizeof(_ffi.Thresholds) == 40
    assert C.sizeof(_ffi.EngineInfo) == 32 + 32 + 64 + 32 + 8


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cutter_vad_amd.core.exceptions import ModelInitializationError
    from cutter_vad_amd.engine import Engine
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    with pytest.raises(ModelInitializationError, match="no HIP device|no CPU fallback|Failed to load model"):
        Engine(blob, max_streams=8)


def test_bad_weights_are_rejected():
    lib = _ffi.lib()
    n = C.c_size_t()
    junk = b"not a blob" * 10
    assert lib.vad_debug_pack_weights(5, junk, len(junk), None, 0, C.byref(n), None) == _ffi.VAD_ERR_BAD_WEIGHTS
    assert b"Failed to load model" in lib.vad_last_create_error()
    with open(weights_io.packaged_blob_path(4), "rb") as f:
        v4 = f.read()
    assert lib.vad_debug_pack_weights(5, v4, len(v4), None, 0, C.byref(n), None) == _ffi.VAD_ERR_BAD_WEIGHTS


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cutter_vad_amd")
    bad = []
    for dp, _dn, fn in os.walk(pkg):
        if "build" in dp.split(os.sep):
            continue
        for f in fn:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|silero_oracle", src, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product code references the oracle: {bad}"


def test_packed_layout_reproduces_the_oracle():
    """NumPy model of the kernel's dataflow over the packed streams == oracle (V5)."""
    from oracle import oracle
    from tests import kernel_model as KM
    from tests.signals import make_streams
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    W, sect = KM.packed_streams(5, blob)
    om = oracle.OracleModel(blob, "f64")
    x = make_streams(32, 3, seed=5)
    hc = np.zeros((32, 256), np.float32)
    hc_o = hc.copy()
    with np.errstate(over="ignore"):
        for t in range(3):
            p, hc = KM.v5_step(W, sect, x[:, t], hc, gate=0.01)
            po = om.step_batch(oracle.denoise(x[:, t]).reshape(32, 512), hc_o)
            assert np.abs(p - po).max() <= 5e-6
            assert np.abs(hc - hc_o).max() <= 1e-4


def test_packed_layout_of_the_v5_8k_submodel_reproduces_the_oracle():
    """The same model of the kernel's dataflow, 8 kHz instantiation (256-sample frames, window 128, 64 + 1 bins)."""
    from oracle import oracle
    from tests import kernel_model as KM
    from tests.signals import make_streams
    with open(weights_io.packaged_blob_path(5, 8000), "rb") as f:
        blob = f.read()
    W, sect = KM.packed_streams(5, blob)
    om = oracle.OracleModel(blob, "f64")
    x = make_streams(32, 2, seed=6).reshape(32, 4, 256)
    hc = np.zeros((32, 256), np.float32)
    hc_o = hc.copy()
    with np.errstate(over="ignore"):
        for t in range(4):
            p, hc = KM.v5_step(W, sect, x[:, t], hc, gate=0.01, k8=True)
            po = om.step_batch(oracle.denoise(x[:, t]).reshape(32, 256), hc_o)
            assert np.abs(p - po).max() <= 5e-6
            assert np.abs(hc - hc_o).max() <= 1e-4


def test_packed_layout_for_the_16_stream_tile_kernel_reproduces_the_oracle():
    """pack_silero_v5_t16 (16 x 16 x 4 MFMA tiles) through the model of silero_v5_t16.hip's dataflow == oracle."""
    from oracle import oracle
    from tests import kernel_model as KM
    from tests.signals import make_streams
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    W, sect = KM.packed_streams(516, blob)
    om = oracle.OracleModel(blob, "f64")
    x = make_streams(16, 3, seed=8)
    hc = np.zeros((16, 256), np.float32)
    hc_o = hc.copy()
    with np.errstate(over="ignore"):
        for t in range(3):
            p, hc = KM.v5_step_t16(W, sect, x[:, t], hc, gate=0.01)
            po = om.step_batch(oracle.denoise(x[:, t]).reshape(16, 512), hc_o)
            assert np.abs(p - po).max() <= 5e-6
            assert np.abs(hc - hc_o).max() <= 1e-4


def test_packed_layout_for_the_16_stream_tile_kernel_8k_sub_model_reproduces_the_oracle():
    """pack_silero_v5_t16 on the 8 kHz blob through the K8 instantiation's dataflow (silero_v5_step16<., ., K8 = true>) == oracle."""
    from oracle import oracle
    from tests import kernel_model as KM
    from tests.signals import make_streams
    with open(weights_io.packaged_blob_path(5, 8000), "rb") as f:
        blob = f.read()
    W, sect = KM.packed_streams(516, blob)
    om = oracle.OracleModel(blob, "f64")
    x = make_streams(16, 2, seed=16).reshape(16, 4, 256)
    hc = np.zeros((16, 256), np.float32)
    hc_o = hc.copy()
    with np.errstate(over="ignore"):
        for t in range(4):
            p, hc = KM.v5_step_t16(W, sect, x[:, t], hc, gate=0.01, k8=True)
            po = om.step_batch(oracle.denoise(x[:, t]).reshape(16, 256), hc_o)
            assert np.abs(p - po).max() <= 5e-6
            assert np.abs(hc - hc_o).max() <= 1e-4


@pytest.mark.parametrize("n_in", [256, 768, 1536])
def test_fused_resampler_operator_packing_reproduces_scipy(n_in):
    """pack_resample_operator_t16 through the model of the fused kernel's resample prologue == scipy.signal.resample."""
    import scipy.signal
    from tests import kernel_model as KM
    W, wave_blocks, row128 = KM.packed_resample_t16(n_in)
    # 24 / 48 kHz: every third sample is a copy, the contraction runs over n_in / 6 folded samples per part instead of n_in / 4
    # ... and of an 8 kHz chunk only the odd output rows are contracted (the even outputs are the input samples)
    assert wave_blocks == (2 + 4 * 4 if n_in == 256 else 4 + 8 * (n_in // 6 // 16))
    x = (0.4 * np.random.default_rng(n_in).standard_normal((16, n_in))).astype(np.float32)
    y = KM.resample_t16(W, wave_blocks, row128, x)
    ref = np.stack([scipy.signal.resample(r.astype(np.float64), 512) for r in x])
    assert np.abs(y - ref).max() <= 2e-6


def test_weights_blob_roundtrip_and_reference_arity_rule(tmp_path):
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    v, t = weights_io.unpack_svw(blob)
    assert v == 5 and t["stft.basis"].shape == (258, 256) and t["lstm.w_ih"].shape == (512, 128)
    assert sum(a.size for a in t.values()) == 309633
    again = weights_io.pack_svw(5, t)
    assert again == blob
    with pytest.raises(weights_io.WeightFormatError):
        weights_io.unpack_svw(b"garbage" * 4)
    p = tmp_path / "w.svw"
    p.write_bytes(blob)
    assert weights_io.load_weight_blob(str(p), 5) == blob
    with pytest.raises(weights_io.WeightFormatError):
        weights_io.load_weight_blob(str(p), 4)


def test_v4_blobs_name_their_graph_branch(tmp_path):
    """V4 ships two sub-models; a blob is tied to its branch by `meta.variant` and the loader refuses the
    wrong one; when the reference's .onnx is at hand the packaged blobs must equal a fresh extraction."""
    import os
    b16 = open(weights_io.packaged_blob_path(4), "rb").read()
    b8 = open(weights_io.packaged_blob_path(4, 48000), "rb").read()
    assert weights_io.packaged_blob_path(4, 8000) == weights_io.packaged_blob_path(4, 24000) != weights_io.packaged_blob_path(4)
    assert weights_io.packaged_blob_path(5, 8000).endswith("silero_v5_8k.svw")             # V5's 8 kHz sub-model (256-sample frames)
    t16, t8 = weights_io.unpack_svw(b16)[1], weights_io.unpack_svw(b8)[1]
    assert "meta.variant" not in t16 and float(t8["meta.variant"][0]) == 8000.0
    assert set(t8) - {"meta.variant"} == set(t16) and not np.array_equal(t8["l0.pw.w"], t16["l0.pw.w"])
    assert np.array_equal(t8["stft.basis"], t16["stft.basis"])          # the DFT basis is shared by the two sub-models
    p16, p8 = tmp_path / "a_16k.svw", tmp_path / "a_8k.svw"
    p16.write_bytes(b16)
    p8.write_bytes(b8)
    assert weights_io.load_weight_blob(str(p8), 4, 24000) == b8
    for path, sr in ((p16, 8000), (p8, 16000)):
        with pytest.raises(weights_io.WeightFormatError, match="sub-model"):
            weights_io.load_weight_blob(str(path), 4, sr)
    onnx = "/root/reference/src/real_time_vad/models/silero_vad.onnx"
    if os.path.exists(onnx):                                             # not on the GPU box
        assert weights_io.load_weight_blob(onnx, 4, 16000) == b16
        assert weights_io.load_weight_blob(onnx, 4, 8000) == b8


def test_resample_operator_matches_scipy_and_reference_fixture():
    """R[512][n_in] applied on the host == scipy.signal.resample == AudioUtils.resample_audio fixture."""
    import scipy.signal
    lib = _ffi.lib()
    g = np.load(os.path.join(ROOT, "tests", "golden", "utils.npz"))
    for sr, n_in in ((8000, 256), (24000, 768), (48000, 1536)):
        R = np.empty((512, n_in), np.float32)
        assert lib.vad_debug_resample_operator(n_in, R.ctypes.data_as(C.POINTER(C.c_float)), R.size) == 0
        x = (0.5 * np.random.default_rng(sr).standard_normal((5, n_in))).astype(np.float32)
        got = (x.astype(np.float64) @ R.astype(np.float64).T).astype(np.float32)
        assert np.abs(got - g[f"resample_{sr}"]).max() <= 2e-6
        live = np.stack([scipy.signal.resample(r, 512).astype(np.float32) for r in x])
        assert np.abs(got - live).max() <= 2e-6
    assert lib.vad_debug_resample_operator(100, None, 0) == _ffi.VAD_ERR_INVALID_ARG


def _build_c_example(tmp_path, name="c_abi_min"):
    import subprocess
    pkg = os.path.join(ROOT, "cutter_vad_amd")
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L", pkg, "-lvad_engine", f"-Wl,-rpath,{pkg}", "-lm", "-o", exe])
    return exe


def test_header_is_plain_c_and_a_c_caller_fails_loudly_without_gpu(tmp_path):
    """include/vad_engine.h compiles as pedantic C99, a C program links against the library through it alone, and -
    in this GPU-less container - vad_engine_create refuses with VAD_ERR_NO_DEVICE instead of computing anything."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    _build_c_example(tmp_path, "c_abi_serve")        # the ABI v2 serving loop compiles as pedantic C99 too
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is covered by tests/test_gpu_v5.py")
    r = subprocess.run([exe, weights_io.packaged_blob_path(5)], capture_output=True, text=True)
    assert r.returncode == 1
    assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


def test_folded_resample_stream_reproduces_scipy():
    """The kernel's folded contraction, evaluated in NumPy from the packed operator stream alone (tests/kernel_model.py),
    equals scipy.signal.resample: guards the packer's layout and the fold algebra without a GPU."""
    import scipy.signal
    from tests import kernel_model
    for n_in in (256, 768, 1536):
        x = (0.5 * np.random.default_rng(n_in).standard_normal((7, n_in))).astype(np.float32)
        got = kernel_model.resample_512(x.astype(np.float64))
        ref = scipy.signal.resample(x.astype(np.float64), 512, axis=1)
        assert np.abs(got - ref).max() <= 1e-6, n_in
    nf, tb, r128 = C.c_size_t(), C.c_uint32(), C.c_uint32()
    assert _ffi.lib().vad_debug_pack_resample(100, None, 0, C.byref(nf), C.byref(tb), C.byref(r128)) == _ffi.VAD_ERR_INVALID_ARG


def test_packed_layout_reproduces_the_oracle_v4():
    """NumPy model of the two V4 launches over the packed streams == oracle."""
    from oracle import oracle
    from tests import kernel_model as KM
    from tests.signals import make_streams
    with open(weights_io.packaged_blob_path(4), "rb") as f:
        blob = f.read()
    W, sect = KM.packed_streams(4, blob)
    om = oracle.OracleModel(blob, "f64")
    x = make_streams(32, 2, seed=6)
    hc = np.zeros((32, 256), np.float32)
    hc_o = hc.copy()
    with np.errstate(over="ignore"):
        for t in range(2):
            p, hc = KM.v4_step(W, sect, x[:, t], hc, gate=0.01)
            po = om.step_batch(oracle.denoise(x[:, t]).reshape(32, 512), hc_o)
            assert np.abs(p - po).max() <= 5e-6
            assert np.abs(hc - hc_o).max() <= 1e-4


@pytest.mark.parametrize("sr", [16000, 8000])
def test_packed_layout_of_the_v4_16_stream_tile_kernel_reproduces_the_oracle(sr):
    """pack_silero_v4_t16's streams, read with silero_v4_t16.hip's indexing (tests/kernel_model.py), == oracle; both sub-models."""
    from oracle import oracle
    from tests import kernel_model as KM
    from tests.signals import make_streams
    with open(weights_io.packaged_blob_path(4, sr), "rb") as f:
        blob = f.read()
    W, sect = KM.packed_streams(416, blob)
    om = oracle.OracleModel(blob, "f64")
    x = make_streams(16, 2, seed=8)
    hc = np.zeros((16, 256), np.float32)
    hc_o = hc.copy()
    with np.errstate(over="ignore"):
        for t in range(2):
            p, hc = KM.v4_step_t16(W, sect, x[:, t], hc, gate=0.01, k8=sr != 16000)
            po = om.step_batch(oracle.denoise(x[:, t]).reshape(16, 512), hc_o)
            assert np.abs(p - po).max() <= 5e-6
            assert np.abs(hc - hc_o).max() <= 1e-4

"""INTEGRATION.md §A, executed (build container only - it needs the reference, which never travels to the GPU box).

The `HipSileroVADModel` stub a maintainer would add next to the reference's `core/silero_model.py` is taken VERBATIM out of the
markdown (so the document cannot drift from what is tested), executed as `real_time_vad.core.hip_silero_model` inside the imported
reference package, and swapped in where the document says: the constructor call at
/root/reference/src/real_time_vad/core/silero_model.py:693.  The library behind its `ctypes.CDLL("libvad_engine.so")` is the REAL
`csrc/engine.cpp` + packers, linked against the test-only HIP stand-in of tools/san_tick/ (host stand-ins for the kernel launches:
p = |first sample of the frame|) - no GPU in the container.  The reference's own `VADWrapper` / `VADProcessor` / state machine then
run on top: scripted probabilities go in through the C ABI, the reference's state machine must fire the scripted events.
A wrong `argtypes` line, struct field or argument order in the document fails here, not at a maintainer's desk.
"""
import ctypes
import os
import re
import shutil
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "real_time_vad")),
                                reason="the reference is only present in the build container")


def _standin_library(tmp) -> str:
    """csrc/engine.cpp as a shared library over the HIP stand-in (tools/san_tick): every C-ABI entry point, no device."""
    from tests import standin
    return standin.build(tmp, "-O1")


def _doc_stub_source() -> str:
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        md = f.read()
    sec = md[md.index("## A. "):md.index("## B. ")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 1 and "class HipSileroVADModel" in blocks[0]
    return blocks[0]


def _import_reference():
    stub = types.ModuleType("onnxruntime")          # the reference imports it at module level; nothing here ever runs a session

    class InferenceSession:
        def __init__(self, *a, **k):
            raise RuntimeError("onnxruntime is not part of this test: the model operator is the HIP engine's")
    stub.InferenceSession = InferenceSession
    stub.SessionOptions = type("SessionOptions", (), {})
    stub.GraphOptimizationLevel = type("GraphOptimizationLevel", (), {"ORT_ENABLE_ALL": 99})
    stub.get_available_providers = lambda: ["CPUExecutionProvider"]
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "onnxruntime" or k.startswith("real_time_vad")}
    sys.modules["onnxruntime"] = stub
    sys.path.insert(0, os.path.join(REF, "src"))
    import real_time_vad
    return real_time_vad, saved


@pytest.fixture()
def reference_with_doc_stub(tmp_path, monkeypatch):
    monkeypatch.setattr(sys, "dont_write_bytecode", True)             # nothing is written under /root/reference
    lib = _standin_library(tmp_path)
    rtv, saved = _import_reference()
    import real_time_vad.core.silero_model as ref_sm
    # the stub module, as the file the document names: src/real_time_vad/core/hip_silero_model.py
    mod = types.ModuleType("real_time_vad.core.hip_silero_model")
    mod.__package__ = "real_time_vad.core"
    real_cdll = ctypes.CDLL
    monkeypatch.setattr(ctypes, "CDLL", lambda name, *a, **k: real_cdll(lib if name == "libvad_engine.so" else name, *a, **k))
    exec(compile(_doc_stub_source(), "INTEGRATION.md#A", "exec"), mod.__dict__)
    monkeypatch.setattr(ctypes, "CDLL", real_cdll)
    sys.modules[mod.__name__] = mod
    # silero_model.py:693: `self.model = HipSileroVADModel(str(model_path), self.config.model_version)`
    monkeypatch.setattr(ref_sm, "SileroVADModel", mod.HipSileroVADModel)
    # the model directory the document's stub expects: <dir>/silero_vad_v5.onnx (the reference checks that it exists) and the
    # .svw blob of tools/extract_weights.py next to it
    models = tmp_path / "models"
    models.mkdir()
    os.symlink(os.path.join(REF, "src/real_time_vad/models/silero_vad_v5.onnx"), models / "silero_vad_v5.onnx")
    shutil.copy(os.path.join(ROOT, "cutter_vad_amd/weights/silero_v5_16k.svw"), models / "silero_vad_v5.svw")
    yield rtv, mod, models
    sys.path.remove(os.path.join(REF, "src"))
    for k in [k for k in sys.modules if k == "onnxruntime" or k.startswith("real_time_vad")]:
        del sys.modules[k]
    sys.modules.update({k: v for k, v in saved.items() if v is not None})


def _frames(probs):
    """a frame whose first sample IS the scripted probability (the stand-in's model); the rest is silence"""
    x = np.zeros((len(probs), 512), np.float32)
    x[:, 0] = probs
    return x


def test_the_documents_stub_runs_under_the_references_own_wrapper(reference_with_doc_stub):
    rtv, mod, models = reference_with_doc_stub
    from real_time_vad import SampleRate, SileroModelVersion, VADConfig, VADWrapper
    cfg = VADConfig(sample_rate=SampleRate.SAMPLERATE_16, model_version=SileroModelVersion.V5, model_path=models,
                    vad_start_probability=0.5, vad_end_probability=0.3, voice_start_frame_count=3, voice_end_frame_count=4,
                    enable_denoising=False, buffer_size=512)
    w = VADWrapper(cfg)
    assert type(w.processor.model) is mod.HipSileroVADModel            # the reference's VADProcessor built the document's class
    events = []
    w.set_callbacks(voice_start_callback=lambda: events.append(("S", len(seen))),
                    voice_end_callback=lambda wav: events.append(("E", len(seen), len(wav))),
                    voice_continue_callback=lambda pcm: None)
    # silence, an utterance of 6 frames above 0.5, silence again: the reference's state machine starts on the third high frame
    # (tests/test_silero_model.py:894-899) and ends on the fourth low one
    script = [0.1, 0.2, 0.9, 0.8, 0.95, 0.7, 0.6, 0.9, 0.2, 0.1, 0.25, 0.05, 0.1, 0.1]
    seen = []
    for fr, p in zip(_frames(script), script):
        w.process_audio_data(fr)
        seen.append(p)
    got = list(w.processor.voice_probabilities)
    assert np.allclose(got, script, atol=1e-7)                          # what came back through vad_step is what went in
    assert [e[:2] for e in events] == [("S", 4), ("E", 11)], events     # START while frame index 4 is processed, END at index 11
    assert events[1][2] > 44                                            # a WAV payload from the reference's own writer
    assert w.processor.model.prediction_count == len(script)
    # reset() reaches vad_stream_reset: the stand-in's "h" counts the frames a stream has seen
    info = (ctypes.c_float * 256)()
    lib = mod._lib
    lib.vad_stream_get_state.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_float)]
    assert lib.vad_stream_get_state(mod._ENGINE, w.processor.model._slot, info) == 0 and info[0] == len(script)
    w.reset()
    assert lib.vad_stream_get_state(mod._ENGINE, w.processor.model._slot, info) == 0 and info[0] == 0
    # a second wrapper shares the engine and gets its own slot (one engine per process, one slot per model object)
    w2 = VADWrapper(cfg)
    assert w2.processor.model._slot.value != w.processor.model._slot.value
    w2.process_audio_data(_frames([0.75])[0])
    assert list(w2.processor.voice_probabilities) == pytest.approx([0.75])
    # short and long chunks are padded / truncated by the stub as the reference's _prepare_audio_input does (:464-468)
    m = w2.processor.model
    assert m.predict(np.full(100, 0.4, np.float32), 16000) == pytest.approx(0.4)
    assert m.predict(np.full(700, 0.3, np.float32), 16000) == pytest.approx(0.3)


def test_a_wrong_argtypes_line_in_the_document_would_fail_this_test(reference_with_doc_stub):
    """The check above is not vacuous: the same stub with vad_step's threshold and format arguments swapped (the mistake a
    drifting document would make) does not produce the scripted probabilities."""
    rtv, mod, models = reference_with_doc_stub
    lib = mod._lib
    good = list(lib.vad_step.argtypes)
    m = mod.HipSileroVADModel(str(models / "silero_vad_v5.onnx"), __import__("real_time_vad").SileroModelVersion.V5)
    assert m.predict(_frames([0.6])[0], 16000) == pytest.approx(0.6)
    try:
        lib.vad_step.argtypes = good[:4] + [ctypes.c_float, ctypes.c_int] + good[6:]
        with pytest.raises(Exception):
            p = m.predict(_frames([0.6])[0], 16000)
            assert p == pytest.approx(0.6)
    finally:
        lib.vad_step.argtypes = good

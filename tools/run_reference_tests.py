#!/usr/bin/env python3
"""Point the REFERENCE's own unit tests at the mirror (build container only; nothing here travels to the GPU box).

SURVEY §4: the reference's tests pin API and behaviour well and numerics not at all - so they are the right judge of the drop-in
surface `cutter_vad_amd.core` / `.utils` keep.  This script runs pytest on /root/reference/tests/test_*.py IN PLACE (the files
are read where they lie, with their own conftest.py; nothing is copied and nothing is written under /root/reference: no
bytecode, no cache, rootdir and ini in a temporary directory), twice, each in its own process:

  * `mirror`:    `real_time_vad*` is an alias of `cutter_vad_amd*` in sys.modules, so `from real_time_vad.core.vad_wrapper import
                 VADWrapper` and `patch('real_time_vad.core.vad_wrapper.VADProcessor')` reach the mirror's modules; the engine behind
                 the default pool is tests/fakes.py's scripted one (no GPU in the container);
  * `reference`: the reference itself from /root/reference/src with a placeholder `onnxruntime` module (the seam its own tests mock,
                 tests/test_silero_model.py:301) - to tell a mirror failure from a test that fails on the reference too.

Every mirror failure is then classified:
  fails-on-reference-too   the same test fails against the reference's own code
  out-of-scope-helper      AttributeError on an offline `AudioUtils` helper that SURVEY §2 row 6 marks out of scope
  needs-gpu                reaches a HIP kernel (resample_audio) - covered by tests/test_gpu_resample.py on the MI355X
  white-box                pokes private members of the reference's implementation (`_state`, `_callbacks`, `_handle_callbacks`,
                           `ort.InferenceSession`, ...) or replaces the processor by a Mock that cannot play the mirror's batched protocol
  PUBLIC-SURFACE           anything else: a bug in the mirror.  The script exits 1 if there is one.

Output: profiles/r04_reference_tests.json (table + every failure with its class and first error line).
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FILES = ("test_config.py", "test_exceptions.py", "test_audio_utils.py", "test_vad_wrapper.py", "test_silero_model.py")
OUT_OF_SCOPE_HELPERS = ("normalize_audio", "apply_window", "calculate_rms", "calculate_energy", "detect_clipping", "load_audio_file",
                        "save_audio_file", "calculate_zero_crossing_rate", "apply_preemphasis", "compute_spectral_centroid")
# what makes a test white-box: it reads or writes private members, patches names inside the reference's modules, substitutes Mock
# processors / models / sessions, or touches the per-frame buffers of the reference's VADProcessor (the mirror keeps them on the device)
WHITE_BOX = (r"\._[a-zA-Z]", r"patch\(\s*'real_time_vad", r"patch\.object\(", r"Mock\(spec=", r"\bort\.",
             r"\bmock_(vad_processor|processor|processor_class|model|silero_model|onnx_session|session|warn)\b",
             r"\.(voice_buffer|current_voice_data|voice_probabilities|wav_writer|recent_voice_start_frames|recent_voice_end_frames|session)\b",
             r"\.is_voice_active\s*=[^=]")


# ----------------------------------------------------------------------------------------------------- child process
def _stub_onnxruntime():
    stub = types.ModuleType("onnxruntime")

    class SessionOptions:
        pass

    class GraphOptimizationLevel:
        ORT_ENABLE_ALL = 99

    class InferenceSession:                      # never runs a model here: the reference's tests replace it by a Mock
        def __init__(self, *a, **k):
            raise RuntimeError("placeholder onnxruntime: no model execution in this harness")

        def get_inputs(self):
            return []

        def get_outputs(self):
            return []

        def get_providers(self):
            return ["CPUExecutionProvider"]

        def run(self, names, feeds):
            raise RuntimeError("placeholder onnxruntime")

    stub.SessionOptions, stub.GraphOptimizationLevel, stub.InferenceSession = SessionOptions, GraphOptimizationLevel, InferenceSession
    stub.get_available_providers = lambda: ["CPUExecutionProvider"]
    sys.modules["onnxruntime"] = stub


def _alias_mirror():
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT)
    import cutter_vad_amd
    from cutter_vad_amd import pool
    from tests.fakes import FakeEngine, FakePool
    for suffix in ("", ".core", ".core.config", ".core.exceptions", ".core.silero_model", ".core.vad_wrapper",
                   ".core.async_vad_wrapper", ".utils", ".utils.audio", ".utils.wav_writer"):
        sys.modules["real_time_vad" + suffix] = importlib.import_module("cutter_vad_amd" + suffix)
    assert sys.modules["real_time_vad"] is cutter_vad_amd

    class HostOnlyPool(FakePool):                # the engine of every wrapper a test builds for real: scripted, deterministic
        def __init__(self):
            super().__init__(FakeEngine(fn=lambda fr: 0.9 if float(np.abs(fr).max()) > 0.3 else 0.05))

        def any_engine(self):
            return self.engine

        def resample(self, chunks, sr_in):
            raise RuntimeError("needs-gpu: the resampler is a HIP kernel (tests/test_gpu_resample.py)")

        def close(self):
            pass
    FakeEngine.resample_generic = lambda self, *a, **k: (_ for _ in ()).throw(
        RuntimeError("needs-gpu: resample_audio is a HIP kernel (tests/test_gpu_resample.py)"))
    pool._default = HostOnlyPool()


def child(target: str, out_path: str) -> int:
    sys.dont_write_bytecode = True
    import pytest
    _stub_onnxruntime()
    if target == "mirror":
        _alias_mirror()
    else:
        sys.path.insert(0, os.path.join(REF, "src"))
        import real_time_vad  # noqa: F401
    tmp = tempfile.mkdtemp(prefix="reftests_")
    ini = os.path.join(tmp, "pytest.ini")
    with open(ini, "w") as f:        # the reference's own options (pyproject.toml:115-126), minus anything that writes files
        f.write("[pytest]\nmarkers =\n    slow: slow\n    integration: integration\n    unit: unit\n")

    class Collect:
        def __init__(self):
            self.rows, self.names = {}, {}

        def pytest_collection_modifyitems(self, items):        # node ids are relative to the (temporary) rootdir: key by file
            for it in items:
                self.names[it.nodeid] = os.path.basename(str(it.fspath)) + "::" + it.nodeid.split("::", 1)[1]

        def pytest_runtest_logreport(self, report):
            row = self.rows.setdefault(self.names.get(report.nodeid, report.nodeid), {"outcome": "passed", "error": ""})
            if report.failed:
                row["outcome"] = "failed" if report.when == "call" else "error"
                row["error"] = _first_error_line(str(report.longrepr))
                row["text"] = str(report.longrepr)[-3000:]
            elif report.skipped and row["outcome"] == "passed":
                row["outcome"] = "skipped"

    c = Collect()
    args = ["-c", ini, "--rootdir", tmp, "-p", "no:cacheprovider", "-q", "--tb=short", "--no-header",
            "-W", "ignore"] + [os.path.join(REF, "tests", f) for f in FILES]
    rc = pytest.main(args, plugins=[c])
    with open(out_path, "w") as f:
        json.dump({"rc": int(rc), "rows": c.rows}, f)
    return 0


def _first_error_line(text: str) -> str:
    lines = [ln for ln in text.splitlines() if ln.startswith("E ")]
    return (lines[0][1:].strip() if lines else text.strip().splitlines()[-1] if text.strip() else "")[:300]


# ----------------------------------------------------------------------------------------------------- parent
def _test_source(nodeid: str) -> str:
    """text of the test function (and of the class fixtures it names) from the reference file, read as text"""
    path, *names = nodeid.split("::")
    fn = names[-1].split("[")[0]
    with open(os.path.join(REF, "tests", os.path.basename(path))) as f:
        src = f.read()
    m = re.search(r"(?:^ *@[^\n]*\n)*^( *)def " + re.escape(fn) + r"\(.*?(?=^\1(?:def |@|class )|\Z)", src, re.S | re.M)
    return m.group(0) if m else ""


def classify(nodeid: str, row: dict, ref_rows: dict) -> str:
    ref = ref_rows.get(nodeid, {}).get("outcome")
    if ref in ("failed", "error"):
        return "fails-on-reference-too"
    err, text = row["error"], row.get("text", "")
    if "AttributeError" in err and any(h in err for h in OUT_OF_SCOPE_HELPERS):
        return "out-of-scope-helper"
    if "needs-gpu" in text or "libvad_engine" in text or "hipErrorNoDevice" in text or "no ROCm-capable device" in text:
        return "needs-gpu"
    src = _test_source(nodeid)
    if any(re.search(p, src) for p in WHITE_BOX):
        return "white-box"
    return "PUBLIC-SURFACE"


def main() -> int:
    if not os.path.isdir(os.path.join(REF, "tests")):
        print("the reference is not here (build container only)")
        return 2
    res = {}
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT)
    for target in ("mirror", "reference"):
        fd, path = tempfile.mkstemp(suffix=".json")
        os.close(fd)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", target, path], env=env, cwd=tempfile.gettempdir(),
                           capture_output=True, text=True)
        try:
            with open(path) as f:
                res[target] = json.load(f)["rows"]
        except (OSError, ValueError):
            print(p.stdout[-3000:], p.stderr[-3000:])
            raise
        finally:
            os.unlink(path)
    table, failures = {}, []
    for f in FILES:
        rows = {k: v for k, v in res["mirror"].items() if os.path.basename(k.split("::")[0]) == f}
        refrows = {k: v for k, v in res["reference"].items() if os.path.basename(k.split("::")[0]) == f}
        entry = {"tests": len(rows), "mirror_passed": sum(v["outcome"] == "passed" for v in rows.values()),
                 "reference_passed": sum(v["outcome"] == "passed" for v in refrows.values()), "mirror_failures_by_class": {}}
        for k, v in sorted(rows.items()):
            if v["outcome"] in ("failed", "error"):
                cls = classify(k, v, res["reference"])
                entry["mirror_failures_by_class"][cls] = entry["mirror_failures_by_class"].get(cls, 0) + 1
                failures.append({"test": k, "class": cls, "error": v["error"]})
        table[f] = entry
    bugs = [x for x in failures if x["class"] == "PUBLIC-SURFACE"]
    out = {"what": "the reference's own unit tests run in place against the mirror (real_time_vad* aliased to cutter_vad_amd*, "
                   "scripted engine) and against the reference itself (placeholder onnxruntime); tools/run_reference_tests.py",
           "table": table, "public_surface_failures": len(bugs), "failures": failures}
    with open(os.path.join(ROOT, "profiles", "r04_reference_tests.json"), "w") as f:
        json.dump(out, f, indent=1)
    for f, e in table.items():
        print(f"{f:24s} mirror {e['mirror_passed']:3d}/{e['tests']:3d}   reference {e['reference_passed']:3d}/{e['tests']:3d}   {e['mirror_failures_by_class']}")
    for b in bugs:
        print("PUBLIC-SURFACE:", b["test"], "|", b["error"])
    return 1 if bugs else 0


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--child":
        sys.exit(child(sys.argv[2], sys.argv[3]))
    sys.exit(main())

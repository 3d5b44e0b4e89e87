"""The reference's own unit tests against the mirror (tools/run_reference_tests.py: in place, aliased imports, scripted engine).
Build container only - the reference does not travel to the GPU box, where this test skips."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="the reference is only present in the build container")
def test_reference_unit_tests_find_no_public_surface_failure_in_the_mirror():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    with open(os.path.join(ROOT, "profiles", "r04_reference_tests.json")) as f:
        res = json.load(f)
    assert res["public_surface_failures"] == 0
    t = res["table"]
    # config and exception trees: every test; utilities: everything but out-of-scope helpers, two tests the reference fails itself
    # and the GPU resampler; wrapper / model files: what does not reach into the reference's private members or mock its internals
    assert t["test_config.py"]["mirror_passed"] == t["test_config.py"]["tests"] == 88
    assert t["test_exceptions.py"]["mirror_passed"] == t["test_exceptions.py"]["tests"] == 56
    assert t["test_audio_utils.py"]["mirror_passed"] >= 11 and t["test_vad_wrapper.py"]["mirror_passed"] >= 52
    assert t["test_silero_model.py"]["mirror_passed"] >= 23
    for f, e in t.items():
        assert set(e["mirror_failures_by_class"]) <= {"out-of-scope-helper", "fails-on-reference-too", "needs-gpu", "white-box"}, (f, e)

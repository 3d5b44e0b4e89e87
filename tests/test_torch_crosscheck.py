"""The third implementation (tools/torch_crosscheck.py): PyTorch operators + an independent protobuf reader over the
reference's .onnx files must reproduce tests/golden/model_*.npz.  Container only (the reference does not travel)."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/root/reference/src/real_time_vad/models/silero_vad_v5.onnx"),
                    reason="the reference's .onnx files exist in the build container only")
def test_torch_operators_reproduce_the_goldens():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "torch_crosscheck.py")], capture_output=True, text=True,
                       timeout=600)
    rows = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert r.returncode == 0, r.stdout + r.stderr
    assert [x["model"] for x in rows] == ["v5_16k", "v5_8k", "v4_16k", "v4_8k"]
    for x in rows:
        assert x["identical"] == x["tensors_compared_with_onnx_lite"] > 200
        assert x["torch_f64_vs_golden_max_dp"] <= 2e-6 and x["torch_f64_vs_golden_max_dstate"] <= 2e-6
        assert x["torch_f32_vs_golden_max_dp"] <= 1e-5

"""The tick assembler's host logic (csrc/engine.cpp: staging, overflow queues, stream open / close under load, segment arena,
copy crew, failing ticks, save / restore of a stream mid-segment) under ThreadSanitizer and AddressSanitizer + UBSan, on the
CPU: tools/san_tick/ compiles the engine against a test-only HIP stand-in (GPU sanitizers are not available on the pool) and
runs 4 producer threads + a stream-churning thread against the ticking main thread.  The same scenario runs against the real
kernels in tests/test_gpu_stress.py."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["thread", "address"])
def test_tick_assembler_is_clean_under(mode):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run([os.path.join(ROOT, "tools", "san_tick", "run.sh"), mode], capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "san_tick: all ok" in out, out[-4000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]

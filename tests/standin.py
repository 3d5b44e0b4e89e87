"""TEST-ONLY: the REAL host side of the engine (csrc/engine.cpp + packers: every C-ABI entry point, the tick assembler, the
segment arena) as a shared library over the HIP stand-in of tools/san_tick/ - host stand-ins for the kernel launches
(p = |first sample of the frame|, the real state machine of csrc/sm_device.h) - for places that have no GPU: the build container's
tests (tests/test_integration_doc.py) and the host-side rehearsal of the serving front (tools/bench_server.py --fake-shards).
Never loaded by the product: cutter_vad_amd/_ffi.py knows one library, the HIP one, and fails without it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["tools/san_tick/fake_kernels.cpp", "cutter_vad_amd/csrc/engine.cpp", "cutter_vad_amd/csrc/pack_weights.cpp",
           "cutter_vad_amd/csrc/resample_tables.cpp"]


def build(out_dir: str, opt: str = "-O2") -> str:
    out = os.path.join(str(out_dir), "libvad_engine.so")
    newest = max(os.path.getmtime(os.path.join(ROOT, s)) for s in SOURCES)
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        os.makedirs(str(out_dir), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", opt, "-fPIC", "-shared", "-Itools/san_tick", "-Icutter_vad_amd/csrc", "-o", out,
                               *SOURCES, "-lpthread"], cwd=ROOT)
    return out

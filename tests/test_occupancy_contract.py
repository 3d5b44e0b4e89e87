"""A performance contract that a compiler or a source change could break without any test noticing: the headline launch (8 192
streams, one frame per call) runs on 16-stream tiles with TWO workgroups per CU (cutter_vad_amd/csrc/engine.cpp: launch();
DESIGN.md 2.1c).  That needs, of the SINGLE-FRAME instantiations of silero_v5_step16, at most 256 registers (two waves per SIMD:
512 / 2) and at most 80 KB of LDS (160 / 2); the multi-frame and fused-resample instantiations are allowed one workgroup per CU.
Checked on the code object's own metadata: the kernel is compiled to assembly for gfx950 (no GPU needed) with the product's flags."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not found")
def test_single_frame_instantiations_of_the_16_stream_kernel_fit_twice_on_a_cu(tmp_path):
    from cutter_vad_amd import _build
    out = tmp_path / "t16.s"
    flags = ["-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-kernarg-preload-count=8"]
    src = open(os.path.join(ROOT, "cutter_vad_amd", "_build.py")).read()
    assert "-amdgpu-mfma-vgpr-form" in src and "-amdgpu-kernarg-preload-count=8" in src      # the flags the product build uses
    subprocess.run([_hipcc(), f"--offload-arch={_build.ARCH}", *flags, "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(ROOT, "cutter_vad_amd", "csrc", "silero_v5_t16.hip")], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.vgpr_count:\s*(\d+)", text, re.S):
        kernels[m.group(2)] = (int(m.group(1)), int(m.group(3)))
    # _Z16silero_v5_step16ILb<F32IN>ELb<RS>ELb<K8>ELb<ONE>EEv...
    one = {k: v for k, v in kernels.items() if re.match(r"_Z16silero_v5_step16ILb[01]ELb0ELb[01]ELb1EE", k)}
    assert len(one) == 4, sorted(kernels)
    for name, (lds, regs) in one.items():
        assert regs <= 256, (name, regs)            # two waves per SIMD
        assert lds <= 80 * 1024, (name, lds)        # two workgroups per CU
    assert all(lds <= 160 * 1024 for lds, _ in kernels.values())

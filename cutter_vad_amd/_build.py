"""Build the HIP engine library in-tree: ``cutter_vad_amd/libvad_engine.so`` (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvad_engine.so")
ARCH = "gfx950"

HIP_SOURCES = ["silero_v5.hip", "silero_v5_t16.hip", "silero_v4.hip", "silero_v4_t16.hip", "resample.hip", "resample_generic.hip", "resample_fft.hip", "vad_util.hip"]
CPP_SOURCES = ["engine.cpp", "pack_weights.cpp", "resample_tables.cpp"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the engine has no non-HIP build")


def sources() -> List[str]:
    out = [os.path.join(CSRC, s) for s in HIP_SOURCES + CPP_SOURCES]
    out += [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    out.append(os.path.join(os.path.dirname(HERE), "include", "vad_engine.h"))
    return out


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in sources())


def build(force: bool = False, verbose: bool = False) -> str:
    # kernel experiments on a GPU box (tools/variants.sh): VAD_KERNEL_DEFINES="-DX -DY" recompiles the .hip files named in
    # VAD_KERNEL_DEFINES_FILES (comma separated; default: all) with those flags and relinks.  Never set by the product.
    import shlex
    defines = shlex.split(os.environ.get("VAD_KERNEL_DEFINES", ""))
    only = [f for f in os.environ.get("VAD_KERNEL_DEFINES_FILES", "").split(",") if f]
    experiment = "VAD_KERNEL_DEFINES" in os.environ
    if not force and not experiment and up_to_date():
        return LIB
    cc = _hipcc()
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    common = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    # MFMA accumulators in VGPRs where they fit: the epilogues (|.|, interpolation, cell) read them with VALU instructions,
    # which cannot address AGPRs - every accumulator kept there costs a v_accvgpr_read/write (measured: V5 49.2 -> 48.9 us)
    kernel_flags = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    # leading scalar kernel arguments arrive in SGPRs with the wave instead of through s_load (gfx950 kernarg preload; the V5
    # kernels have such arguments: 0.15 - 0.25 us of 50 per headline step)
    preload = ["-mllvm", "-amdgpu-kernarg-preload-count=8"]
    per_file_flags = {"silero_v5.hip": preload, "silero_v5_t16.hip": preload}
    for s in HIP_SOURCES:
        o = os.path.join(bdir, s + ".o")
        mine = experiment and (not only or s in only)
        if experiment and not mine and os.path.exists(o) and not force:
            objs.append(o)                      # an experiment rebuilds only the files it names
            continue
        cmd = [cc, f"--offload-arch={ARCH}", *common, *kernel_flags, *per_file_flags.get(s, []), *(defines if mine else []), "-c",
               os.path.join(CSRC, s), "-o", o]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
        objs.append(o)
    for s in CPP_SOURCES:
        o = os.path.join(bdir, s + ".o")
        if experiment and os.path.exists(o) and not force:
            objs.append(o)
            continue
        subprocess.check_call([cc, *common, "-fvisibility=hidden", "-c", os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    tmp = LIB + ".tmp"
    subprocess.check_call([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs, "-lpthread"])
    os.replace(tmp, LIB)
    return LIB


def wirebox_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_wirebox" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_wirebox(force: bool = False) -> str:
    """The serving pool's frame inbox (csrc/wirebox.c), a plain-C CPython extension: host plumbing, optional - the pool queues
    frames in Python when it is absent (same results, ~3x the per-frame cost)."""
    import sysconfig
    src, out = os.path.join(CSRC, "wirebox.c"), wirebox_path()
    if not force and os.path.exists(out) and os.path.getmtime(src) <= os.path.getmtime(out):
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    inc = sysconfig.get_paths()["include"]
    if not cc or not os.path.exists(os.path.join(inc, "Python.h")):
        raise RuntimeError("a C compiler and Python.h are needed to build _wirebox")
    tmp = out + ".tmp"
    subprocess.check_call([cc, "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-missing-field-initializers", "-Wno-cast-function-type",
                           f"-I{inc}", f"-I{os.path.join(os.path.dirname(HERE), 'include')}", src, "-o", tmp, "-lpthread"])
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_wirebox(force="--force" in sys.argv))

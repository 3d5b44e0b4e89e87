#!/usr/bin/env python3
"""Silero V4 on the period-64 full-scale square wave (exact nulls in complex spectral bins under log(1 + |X| 2^20)), 8 frames:
the CONDITIONING of the float64 oracle at float32 input precision (tests/signals.py::one_ulp_conditioning: max |dp| under a
one-ulp relative perturbation of the input samples, 256 seeded trials) - the bar of tests/test_gpu_v4.py on this input - next to
what float32 evaluations show against the float64 oracle: the oracle's own float32 build (`--cpu`), PyTorch's float32 operators
(recorded in profiles/r03_f32_yardsticks.json), the kernels on both tile shapes (`--gpu`, MI355X).
`--cpu` -> profiles/r04_v4_square_conditioning_cpu.json, `--gpu` -> gpurun_out/r04_v4_square_conditioning_gpu.json, `--merge` joins
them into profiles/r04_v4_square_conditioning.json.    TEST INFRASTRUCTURE: imports oracle/."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutter_vad_amd import weights_io  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.signals import one_ulp_conditioning  # noqa: E402

SQ = np.where(np.arange(512 * 8) % 64 < 32, 1.0, -1.0).astype(np.float32).reshape(8, 512)


def blob(sr):
    with open(weights_io.packaged_blob_path(4, sr), "rb") as f:
        return f.read()


def run(om, fr):
    n, T, _ = fr.shape
    st = np.zeros((n, 256), np.float32)
    out = np.empty((n, T), np.float32)
    for t in range(T):
        out[:, t] = om.step_batch(oracle.denoise(np.ascontiguousarray(fr[:, t]), 0.01).reshape(n, 512), st, nthreads=os.cpu_count() or 8)
    return out


def cpu():
    res = {}
    for sr in (16000, 8000):
        o64, o32 = oracle.OracleModel(blob(sr), "f64"), oracle.OracleModel(blob(sr), "f32")
        p64 = run(o64, SQ[None])[0]
        res[str(sr)] = {"one_ulp_conditioning": one_ulp_conditioning(lambda fr: run(o64, fr), SQ),
                        "half_ulp_conditioning": float(np.abs(run(o64, (SQ[None].astype(np.float64) * (1 + np.random.default_rng(0).uniform(
                            -2.0 ** -24, 2.0 ** -24, (256, 8, 512)))).astype(np.float32)) - p64[None]).max()),
                        "oracle_f32_max_dp": float(np.abs(run(o32, SQ[None])[0] - p64).max()), "p_f64": p64.tolist()}
    with open(os.path.join(ROOT, "profiles", "r04_v4_square_conditioning_cpu.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def gpu():
    from cutter_vad_amd.engine import Engine
    res = {}
    for sr in (16000, 8000):
        p64 = run(oracle.OracleModel(blob(sr), "f64"), SQ[None])[0]
        for tile in (16, 32):
            with Engine(blob(sr), model_version=4, max_streams=64, sample_rate=sr) as e:
                e.set_tile(tile)
                s = e.open_stream()
                got = np.array([e.step([s], f[None])[0] for f in SQ], np.float32)
            res.setdefault(str(sr), {})[f"kernel_tile{tile}_max_dp"] = float(np.abs(got - p64).max())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_v4_square_conditioning_gpu.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def merge():
    with open(os.path.join(ROOT, "profiles", "r04_v4_square_conditioning_cpu.json")) as f:
        c = json.load(f)
    with open(os.path.join(ROOT, "gpurun_out", "r04_v4_square_conditioning_gpu.json")) as f:
        g = json.load(f)
    with open(os.path.join(ROOT, "profiles", "r03_f32_yardsticks.json")) as f:
        y = json.load(f)
    out = {"what": __doc__.strip().split("\n`--cpu`")[0]}
    for sr in ("16000", "8000"):
        r3 = y["cpu"][f"v4_{sr}_square"]
        out[sr] = {**{k: v for k, v in c[sr].items() if k != "p_f64"}, **g[sr], "torch_f32_max_dp_round3": max(r3.get("torch_f32_abs_dp", [0.0])),
                   "kernel_tile16_max_dp_round3_before_the_second_fold": max(y["gpu"][f"v4_{sr}_square_tile16"]["kernel_abs_dp"])}
    with open(os.path.join(ROOT, "profiles", "r04_v4_square_conditioning.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"--gpu": gpu, "--merge": merge}.get(sys.argv[1] if len(sys.argv) > 1 else "--cpu", cpu)()

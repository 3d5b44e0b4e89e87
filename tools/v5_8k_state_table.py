#!/usr/bin/env python3
"""Silero V5 8 kHz sub-model, the six golden cases of tests/test_gpu_v5_8k.py: final (h, c) relative to the float64 interpreter
golden, |d| / max(1, |ref|), and max |dp| - for the kernel (`--gpu`, MI355X), the oracle's own float32 build and PyTorch's float32
operators (`--cpu`, build container: PyTorch's leg reads the reference's .onnx through tools/torch_crosscheck.py) - plus the
consequence check of that test: 200 further frames from each evaluation's own final state, max |dp| against the float64 oracle
continuing from ITS state.

The test's state bar is the FIXED number STATE_REL = 2e-4; this table documents where the three evaluations sit under it and is
not an input of any test.  `--cpu` writes profiles/r04_v5_8k_state_table_cpu.json, `--gpu` writes gpurun_out/r04_v5_8k_state_table_gpu.json;
`--merge` joins the two into profiles/r04_v5_8k_state_table.json.   TEST INFRASTRUCTURE: imports oracle/.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutter_vad_amd import weights_io  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.signals import model_cases_8k  # noqa: E402
from tests.test_gpu_v5_8k import STATE_REL, continuation_8k  # noqa: E402

NAMES = ("speech_gate", "noise_0.3", "harmonic", "zeros", "square_fullscale", "short200_padded")
GOLD = os.path.join(ROOT, "tests", "golden")


def _inputs():
    with open(weights_io.packaged_blob_path(5, 8000), "rb") as f:
        blob = f.read()
    g = np.load(os.path.join(GOLD, "model_v5_8k.npz"))
    pcm = np.load(os.path.join(GOLD, "speech16k_i16.npz"))["pcm"]
    return blob, g, model_cases_8k(pcm), continuation_8k()


def _rel(a, ref):
    return float((np.abs(a - ref) / np.maximum(1.0, np.abs(ref))).max())


def _oracle_run(om, frames, st):
    return np.array([om.step_batch(np.ascontiguousarray(f[None]), st, nthreads=1)[0] for f in frames], np.float32)


def cpu():
    blob, g, cases, cont = _inputs()
    o64, o32 = oracle.OracleModel(blob, "f64"), oracle.OracleModel(blob, "f32")
    torch_models = None
    try:
        import torch
        from tools import torch_crosscheck as tc
        torch.set_num_threads(4)
        tensors, _ = tc.walk(os.path.join(tc.MODELS, "silero_vad_v5.onnx"))
        torch_models = tc.TorchV5(tensors, torch.float32, k8=True)
    except (FileNotFoundError, ImportError):
        pass
    out = {}
    for name in NAMES:
        ref_s, ref_p = g[f"{name}.state"].reshape(-1), g[f"{name}.probs"]
        s64, s32 = np.zeros((1, 256), np.float32), np.zeros((1, 256), np.float32)
        _oracle_run(o64, cases[name], s64)
        p32 = _oracle_run(o32, cases[name], s32)
        row = {"frames": int(len(cases[name])), "max_abs_ref_state": float(np.abs(ref_s).max()),
               "oracle_f64_rel_dstate": _rel(s64[0], ref_s),
               "oracle_f32_rel_dstate": _rel(s32[0], ref_s), "oracle_f32_max_dp": float(np.abs(p32 - ref_p).max())}
        c64 = _oracle_run(o64, cont, s64)
        row["oracle_f32_continuation_max_dp"] = float(np.abs(_oracle_run(o32, cont, s32) - c64).max())
        if torch_models is not None:
            both = np.concatenate([cases[name], cont])
            tp, _ = torch_models.run(both)
            tp_case, ts = torch_models.run(cases[name])
            row["torch_f32_rel_dstate"] = _rel(np.asarray(ts).reshape(-1), ref_s)
            row["torch_f32_max_dp"] = float(np.abs(np.asarray(tp_case).reshape(-1) - ref_p).max())
            row["torch_f32_continuation_max_dp"] = float(np.abs(np.asarray(tp).reshape(-1)[len(cases[name]):] - c64).max())
        out[name] = row
    path = os.path.join(ROOT, "profiles", "r04_v5_8k_state_table_cpu.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def gpu():
    from cutter_vad_amd.engine import Engine
    blob, g, cases, cont = _inputs()
    o64 = oracle.OracleModel(blob, "f64")
    out = {}
    with Engine(blob, model_version=5, max_streams=64, sample_rate=8000) as eng:
        s = eng.open_stream()
        for name in NAMES:
            ref_s, ref_p = g[f"{name}.state"].reshape(-1), g[f"{name}.probs"]
            eng.reset([s])
            got = np.array([eng.step([s], f[None], denoise=None)[0] for f in cases[name]], np.float32)
            s64 = np.zeros((1, 256), np.float32)
            _oracle_run(o64, cases[name], s64)
            row = {"kernel_rel_dstate": _rel(eng.get_state(s), ref_s), "kernel_max_dp": float(np.abs(got - ref_p).max())}
            gc = np.array([eng.step([s], f[None], denoise=None)[0] for f in cont], np.float32)
            row["kernel_continuation_max_dp"] = float(np.abs(gc - _oracle_run(o64, cont, s64)).max())
            out[name] = row
        out["_device"] = eng.info()["device_name"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_v5_8k_state_table_gpu.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def merge():
    with open(os.path.join(ROOT, "profiles", "r04_v5_8k_state_table_cpu.json")) as f:
        c = json.load(f)
    with open(os.path.join(ROOT, "gpurun_out", "r04_v5_8k_state_table_gpu.json")) as f:
        k = json.load(f)
    res = {"what": __doc__.strip().split("\n\n")[0], "state_bar_in_the_test": STATE_REL, "probability_bar_in_the_test": 2e-5,
           "device": k.pop("_device", None), "cases": {n: {**c[n], **k[n]} for n in NAMES}}
    worst = {key: max(v.get(key, 0.0) for v in res["cases"].values())
             for key in ("kernel_rel_dstate", "oracle_f32_rel_dstate", "torch_f32_rel_dstate", "kernel_continuation_max_dp")}
    res["largest_over_the_six_cases"] = worst
    with open(os.path.join(ROOT, "profiles", "r04_v5_8k_state_table.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(worst))


if __name__ == "__main__":
    {"--gpu": gpu, "--merge": merge}.get(sys.argv[1] if len(sys.argv) > 1 else "--cpu", cpu)()

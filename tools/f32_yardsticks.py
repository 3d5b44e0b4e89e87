#!/usr/bin/env python3
"""float32 yardsticks for the two parity bars that are not the flat 2e-5 / 2e-4 of every other test:

  * Silero V4 on the period-64 full-scale square wave (exact nulls in COMPLEX spectral bins under log(1 + |X| 2^20)):
    |dp| per frame vs the float64 oracle for (a) the oracle's own float32 build, (b) PyTorch's float32 operators
    (tools/torch_crosscheck.py, container only: reads the reference's .onnx), (c) the kernels (both tile shapes) - `--gpu`;
  * Silero V5 8 kHz, final (h, c) after the 529-frame speech golden: the same three against the float64 state.

`--cpu` writes gpurun_out/r03_f32_yardsticks_cpu.json (container), `--gpu` writes gpurun_out/r03_f32_yardsticks_gpu.json (MI355X);
tests/test_gpu_v4.py and tests/test_gpu_v5_8k.py take their bars from the oracle's float32 build, live, on the same input.
TEST INFRASTRUCTURE: imports oracle/.
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

T_SQ = 8


def square(T=T_SQ, L=512):
    return np.where(np.arange(L * T) % 64 < 32, 1.0, -1.0).astype(np.float32).reshape(T, L)


def oracle_run(om, frames):
    st = np.zeros((1, 256), np.float32)
    out = [float(om.step_batch(np.ascontiguousarray(f[None]), st, nthreads=1)[0]) for f in frames]
    return np.array(out, np.float32), st[0].copy()


def blob(version, sr):
    with open(weights_io.packaged_blob_path(version, sr), "rb") as f:
        return f.read()


def speech8k():
    pcm = np.load(os.path.join(ROOT, "tests/golden/speech16k_i16.npz"))["pcm"]
    return model_cases_8k(pcm)["speech_gate"]


def cpu():
    res = {}
    for sr in (16000, 8000):
        o64, o32 = oracle.OracleModel(blob(4, sr), "f64"), oracle.OracleModel(blob(4, sr), "f32")
        fr = square(T_SQ, 512)
        p64, _ = oracle_run(o64, fr)
        p32, _ = oracle_run(o32, fr)
        res[f"v4_{sr}_square"] = {"p_f64": p64.tolist(), "oracle_f32_abs_dp": np.abs(p32 - p64).tolist()}
    o64, o32 = oracle.OracleModel(blob(5, 8000), "f64"), oracle.OracleModel(blob(5, 8000), "f32")
    sp = speech8k()
    _, s64 = oracle_run(o64, sp)
    _, s32 = oracle_run(o32, sp)
    rel = np.abs(s32 - s64) / np.maximum(1.0, np.abs(s64))
    res["v5_8000_speech_state"] = {"frames": int(sp.shape[0]), "max_abs_state_f64": float(np.abs(s64).max()),
                                   "oracle_f32_max_abs_dstate": float(np.abs(s32 - s64).max()), "oracle_f32_max_rel_dstate": float(rel.max())}
    try:        # PyTorch's operators on the reference's own files: container only
        import torch
        from tools import torch_crosscheck as tc
        torch.set_num_threads(4)
        tensors, nodes = tc.walk(os.path.join(tc.MODELS, "silero_vad.onnx"))
        for sr, k8 in ((16000, False), (8000, True)):
            m64, m32 = tc.TorchV4(tensors, nodes, torch.float64, k8=k8), tc.TorchV4(tensors, nodes, torch.float32, k8=k8)
            fr = square(T_SQ, 512)
            p64, _ = m64.run(fr)
            p32, _ = m32.run(fr)
            res[f"v4_{sr}_square"]["torch_f32_abs_dp"] = np.abs(p32 - p64).tolist()
            res[f"v4_{sr}_square"]["torch_f64_vs_oracle_f64_max_dp"] = float(np.abs(p64 - np.array(res[f"v4_{sr}_square"]["p_f64"])).max())
        tensors, _ = tc.walk(os.path.join(tc.MODELS, "silero_vad_v5.onnx"))
        m64, m32 = tc.TorchV5(tensors, torch.float64, k8=True), tc.TorchV5(tensors, torch.float32, k8=True)
        _, t64 = m64.run(sp)
        _, t32 = m32.run(sp)
        res["v5_8000_speech_state"]["torch_f32_max_abs_dstate"] = float(np.abs(t32 - t64).max())
        res["v5_8000_speech_state"]["torch_f32_max_rel_dstate"] = float((np.abs(t32 - t64) / np.maximum(1.0, np.abs(t64))).max())
    except FileNotFoundError:
        pass
    out = os.path.join(ROOT, "gpurun_out", "r03_f32_yardsticks_cpu.json")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def gpu():
    from cutter_vad_amd.engine import Engine
    res = {}
    for sr in (16000, 8000):
        o64 = oracle.OracleModel(blob(4, sr), "f64")
        fr = square(T_SQ, 512)
        p64, _ = oracle_run(o64, fr)
        for tile in (16, 32):
            with Engine(blob(4, sr), model_version=4, max_streams=64, sample_rate=sr) as e:
                e.set_tile(tile)
                s = e.open_stream()
                got = np.array([e.step([s], f[None], denoise=None)[0] for f in fr], np.float32)
            res[f"v4_{sr}_square_tile{tile}"] = {"kernel_abs_dp": np.abs(got - p64).tolist()}
    o64 = oracle.OracleModel(blob(5, 8000), "f64")
    sp = speech8k()
    _, s64 = oracle_run(o64, sp)
    with Engine(blob(5, 8000), model_version=5, max_streams=64, sample_rate=8000) as e:
        s = e.open_stream()
        for f in sp:
            e.step([s], f[None], denoise=None)
        st = e.get_state(s)
    res["v5_8000_speech_state"] = {"kernel_max_abs_dstate": float(np.abs(st - s64).max()),
                                   "kernel_max_rel_dstate": float((np.abs(st - s64) / np.maximum(1.0, np.abs(s64))).max())}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_f32_yardsticks_gpu.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if "--gpu" in sys.argv:
        gpu()
    else:
        cpu()

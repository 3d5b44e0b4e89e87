#!/usr/bin/env python3
"""|dp| statistics over a large sample, per model: the HIP kernel against the f64 oracle, and - as the yardstick for what
ANY float32 evaluation of the same graph does - the oracle's own float32 build (dense, sequential sums) against the same
f64 oracle on the same frames.  All three run free from zero state (a deviation is carried on by the LSTM, as in a stream).

    python3 tools/parity_sweep.py [streams=4096] [frames=256] [out.json]      (GPU box; 4096 x 256 = 1 M frames)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutter_vad_amd import weights_io  # noqa: E402
from cutter_vad_amd.engine import Engine  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.signals import make_streams  # noqa: E402


def stats(d):
    return {"max": float(d.max()), "mean": float(d.mean()), "p99": float(np.percentile(d, 99)),
            "p99.99": float(np.percentile(d, 99.99)), "over_2e-5": int((d > 2e-5).sum()),
            "over_5e-5": int((d > 5e-5).sum()), "over_1e-4": int((d > 1e-4).sum())}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "parity_sweep.json")
    frames = make_streams(n, T, seed=4242)
    nthreads = min(16, os.cpu_count() or 1)
    res = []
    # (model, sr, forced tile shape): V5 on both tile shapes; V5's 8 kHz sub-model sees the same audio as 256-sample frames
    for v, sr, tile in ((5, 16000, 32), (5, 16000, 16), (5, 8000, 32), (5, 8000, 16), (4, 16000, 16), (4, 16000, 32), (4, 8000, 16), (4, 8000, 32)):
        blob = open(weights_io.packaged_blob_path(v, sr), "rb").read()
        o64, o32 = oracle.OracleModel(blob, "f64"), oracle.OracleModel(blob, "f32")
        L = o64.frame_samples
        fr = frames if L == 512 else frames.reshape(n, -1, 256)[:, :T]
        with Engine(blob, model_version=v, max_streams=n, sample_rate=sr) as eng:
            if tile:
                eng.set_tile(tile)
            slots = eng.open_streams(n)
            s64, s32 = np.zeros((n, 256), np.float32), np.zeros((n, 256), np.float32)
            dk, do = np.empty((n, T)), np.empty((n, T))
            for t in range(T):
                x = oracle.denoise(fr[:, t]).reshape(n, L)
                got = eng.step(slots, np.ascontiguousarray(fr[:, t]))
                ref = om_step(o64, x, s64, nthreads)
                r32 = om_step(o32, x, s32, nthreads)
                dk[:, t] = np.abs(got.astype(np.float64) - ref)
                do[:, t] = np.abs(r32.astype(np.float64) - ref)
                if t % 32 == 31:
                    print(f"v{v}/{sr}/tile{tile}: frame {t + 1}/{T}", flush=True)
        worst = np.dstack(np.unravel_index(np.argsort(dk, axis=None)[::-1][:8], dk.shape))[0]
        row = {"version": v, "sample_rate": sr, "streams_per_tile": tile or 32, "frame_samples": L, "frames": n * T,
               "hip_vs_f64": stats(dk), "oracle_f32_vs_f64": stats(do),
               "hip_worst": [{"stream": int(i), "frame": int(t), "dp": float(dk[i, t])} for i, t in worst]}
        res.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


def om_step(om, x, st, nthreads):
    return om.step_batch(x, st, nthreads=nthreads).astype(np.float64)


if __name__ == "__main__":
    main()

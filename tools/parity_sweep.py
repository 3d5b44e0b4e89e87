#!/usr/bin/env python3
"""max / mean / percentiles of |dp| (HIP vs the f64 oracle) over a large sample, per model version (GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutter_vad_amd import weights_io  # noqa: E402
from cutter_vad_amd.engine import Engine  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.signals import make_streams  # noqa: E402

if __name__ == "__main__":
    n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 24
    frames = make_streams(n, T, seed=4242)
    for v in (5, 4):
        blob = open(weights_io.packaged_blob_path(v), "rb").read()
        om = oracle.OracleModel(blob, "f64")
        with Engine(blob, model_version=v, max_streams=n) as eng:
            slots = eng.open_streams(n)
            st = np.zeros((n, 256), np.float32)
            d = np.empty((n, T))
            for t in range(T):
                got = eng.step(slots, frames[:, t])
                ref = om.step_batch(oracle.denoise(frames[:, t]).reshape(n, 512), st, nthreads=16)
                d[:, t] = np.abs(got.astype(np.float64) - ref)
        worst = np.dstack(np.unravel_index(np.argsort(d, axis=None)[::-1][:24], d.shape))[0]
        extra = [{"stream": int(i), "frame": int(t), "dp": float(d[i, t])} for i, t in worst]
        json.dump(extra, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                           f"parity_worst_v{v}.json"), "w"))
        print(json.dumps({"version": v, "frames": n * T, "max": d.max(), "mean": d.mean(),
                          "p99": float(np.percentile(d, 99)), "p99.99": float(np.percentile(d, 99.99)),
                          "over_2e-5": int((d > 2e-5).sum()), "over_5e-5": int((d > 5e-5).sum())}), flush=True)

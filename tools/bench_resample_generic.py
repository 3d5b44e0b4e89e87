#!/usr/bin/env python3
"""Wall time of AudioUtils.resample_audio's generic kernel (vad_resample_generic, host buffers in and out) next to scipy on the
box's host, one JSON line per shape.  `entries` = n_in * n_out operator entries evaluated per array."""
import json
import os
import sys
import time

import numpy as np
import scipy.signal

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutter_vad_amd import weights_io  # noqa: E402
from cutter_vad_amd.engine import Engine  # noqa: E402

SHAPES = [(1, 100, 50), (1, 4608, 1536), (1, 48000, 16000), (1, 44100, 16000), (1, 30011, 10007), (64, 1411, 512),
          (1, 480000, 160000), (1, 441000, 160000), (2, 1440000, 480000), (1, 2880000, 960000), (1, 26460000, 9600000)]
PATHS = {0: "by size", 1: "direct", 2: "fft"}


def main():
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        eng = Engine(f.read(), max_streams=256)
    rng = np.random.default_rng(1)
    for rows, n_in, n_out in SHAPES:
      x = (0.3 * rng.standard_normal((rows, n_in))).astype(np.float32)
      for mode in (1, 2):
        if mode == 1 and rows * n_in * n_out > 2 ** 38:
            continue                                           # the direct kernel would take seconds
        eng.set_resample_path(mode)
        t0 = time.perf_counter()
        y = eng.resample_generic(x, n_out)                 # builds + uploads the tables
        first = time.perf_counter() - t0
        reps = []
        for _ in range(5):                                     # median: the second call of a new large shape was seen to take ~8 ms once
            t0 = time.perf_counter()
            y = eng.resample_generic(x, n_out)
            reps.append(time.perf_counter() - t0)
        warm = sorted(reps)[len(reps) // 2]
        t0 = time.perf_counter()
        ref = scipy.signal.resample(x, n_out, axis=1).astype(np.float32)
        cpu = time.perf_counter() - t0
        print(json.dumps({"path": PATHS[mode], "rows": rows, "n_in": n_in, "n_out": n_out, "entries": rows * n_in * n_out, "first_call_ms": round(first * 1e3, 3),
                          "warm_call_ms": round(warm * 1e3, 3), "slowest_warm_call_ms": round(max(reps) * 1e3, 3), "entries_per_s": round(rows * n_in * n_out / warm, 0),
                          "scipy_host_ms": round(cpu * 1e3, 3), "max_abs_diff_vs_scipy": float(np.abs(y - ref).max())}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()

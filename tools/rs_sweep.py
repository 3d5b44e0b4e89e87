"""Resampler launch-shape sweep: us per launch against chunks per launch (48 kHz), to see what workgroup co-residency buys."""
import json
import sys

import torch

sys.path.insert(0, ".")
from cutter_vad_amd import weights_io  # noqa: E402
from cutter_vad_amd.engine import Engine  # noqa: E402
from tools.bench_configs import timed  # noqa: E402

eng = Engine(open(weights_io.packaged_blob_path(5), "rb").read(), max_streams=64)
import numpy as np  # noqa: E402
import scipy.signal  # noqa: E402

for sr, n_in in ((48000, 1536), (24000, 768), (8000, 256)):
    xs = (0.3 * np.random.default_rng(sr).standard_normal((2080, n_in))).astype(np.float32)   # 65 tiles: both launch shapes below
    for nrows in (2080, 64 * 260):
        xx = np.resize(xs, (nrows, n_in))
        got = eng.resample(xx, sr)
        ref = scipy.signal.resample(xx.astype(np.float64), 512, axis=1)
        print(json.dumps({"sr": sr, "chunks": nrows, "max_abs_err_vs_scipy_f64": float(np.abs(got - ref).max())}), flush=True)
ts = torch.cuda.Stream()
for sr, n_in in ((48000, 1536), (24000, 768), (8000, 256)):
    for B in (1024, 2048, 4096, 8192, 16384, 32768):
        x = (0.1 * torch.randn(2, B, n_in, device="cuda")).contiguous()
        y = torch.empty(B, 512, device="cuda")

        def step(i):
            assert eng._lib.vad_resample_device(eng.handle, x[i % 2].data_ptr(), B, n_in, sr, y.data_ptr(), ts.cuda_stream) == 0

        dt = timed(step, [ts])
        print(json.dumps({"sr": sr, "chunks": B, "us": round(dt * 1e6, 2), "ns_per_chunk": round(dt * 1e9 / B, 3),
                          "executed_frac": round(2.0 * 128 * n_in * B / dt / 157.3e12, 3)}), flush=True)
eng.close()

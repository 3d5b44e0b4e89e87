#!/usr/bin/env python3
"""|dp| statistics of the fused resample -> step launch (vad_step_rates, configs[3]) over a large sample: 4 096 streams in the
stated thirds (1 365 at 8 kHz, 1 365 at 24 kHz, 1 366 at 48 kHz: 256 sixteen-stream tiles, two of them straddling a rate
boundary) for T ticks, free-running from zero state, against the oracle chain Fourier resample (scipy.signal.resample in float64
= oracle.resample to 1e-13, but O(n log n)) -> gate -> float64 model.  Per-rate statistics and those of the two boundary tiles.

    python3 tools/parity_sweep_rates.py [ticks=64] [out.json]      (GPU box)
"""
import json
import os
import sys

import numpy as np
import scipy.signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutter_vad_amd import weights_io  # noqa: E402
from cutter_vad_amd.engine import Engine  # noqa: E402
from oracle import oracle  # noqa: E402
from tools.parity_sweep import stats  # noqa: E402


def synth(n, samples, sr, seed):
    """the bench generator's two kinds of stream, written at `sr`"""
    t = np.arange(samples, dtype=np.float64) / sr
    out = np.empty((n, samples), np.float32)
    for i in range(n):
        rng = np.random.default_rng(seed + i)
        if i % 2 == 0:
            x = 0.02 * rng.standard_normal(samples)
        else:
            x = 0.3 * (0.4 * np.sin(2 * np.pi * 150 * t) + 0.3 * np.sin(2 * np.pi * 300 * t) + 0.2 * np.sin(2 * np.pi * 600 * t)) \
                + 0.03 * rng.standard_normal(samples)
        out[i] = np.clip(x, -1, 1)
    return out


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "parity_sweep_rates.json")
    rates = ((8000, 256, 1365), (24000, 768, 1365), (48000, 1536, 1366))
    B = sum(r[2] for r in rates)
    starts = np.cumsum([0] + [r[2] for r in rates])
    audio = [synth(cnt, T * n_in, sr, 9000 + 10000 * k) for k, (sr, n_in, cnt) in enumerate(rates)]
    blob = open(weights_io.packaged_blob_path(5), "rb").read()
    nth = min(16, os.cpu_count() or 1)
    results = {}
    # The gate (x if |x| > 0.01 else 0, utils/audio.py:117-118) is a step function of the RESAMPLED sample: a value within
    # rounding of the threshold may fall on either side in two float32 evaluations of the same resampler, which moves that
    # sample by 0.01 and the probability with it.  So: gate off measures the arithmetic; gate on is reported next to the same
    # deviation of scipy's OWN float32 transform (what the reference computes for float32 input) from the float64 chain.
    for gate in (None, 0.01):
        o64, y64 = oracle.OracleModel(blob, "f64"), oracle.OracleModel(blob, "f64")
        st, sty = np.zeros((B, 256), np.float32), np.zeros((B, 256), np.float32)
        d, dy = np.empty((B, T)), np.empty((B, T))
        near = 0
        with Engine(blob, model_version=5, max_streams=B) as eng:
            slots = eng.open_streams(B)
            for t in range(T):
                segs = [(np.ascontiguousarray(audio[k][:, t * n_in:(t + 1) * n_in]), sr) for k, (sr, n_in, _) in enumerate(rates)]
                launches = eng.info()["steps"]
                p, _ev, _seg = eng.step_rates(segs, slots, denoise=gate)
                assert eng.info()["steps"] == launches + 1                 # ONE launch
                x16 = np.concatenate([scipy.signal.resample(a.astype(np.float64), 512, axis=1) for a, _ in segs]).astype(np.float32)
                x32 = np.concatenate([scipy.signal.resample(a, 512, axis=1) for a, _ in segs]).astype(np.float32)   # float32 transform
                if gate is not None:
                    near += int((np.abs(np.abs(x16) - np.float32(gate)) < 2e-6).sum())
                    x16, x32 = oracle.denoise(x16, gate).reshape(B, 512), oracle.denoise(x32, gate).reshape(B, 512)
                ref = o64.step_batch(np.ascontiguousarray(x16), st, nthreads=nth)
                yard = y64.step_batch(np.ascontiguousarray(x32), sty, nthreads=nth)
                d[:, t] = np.abs(p.astype(np.float64) - ref)
                dy[:, t] = np.abs(yard.astype(np.float64) - ref)
                if t % 16 == 15:
                    print(f"gate {gate}: tick {t + 1}/{T}", flush=True)
        results[gate] = (d, dy, near)
    # walk order inside the launch: 48 k | 8 k | 24 k -> boundary tiles = (last 6 of 48 k + first 10 of 8 k), (last 3 of 8 k + first 13 of 24 k)
    n48, n8 = rates[2][2], rates[0][2]
    t1 = np.concatenate([np.arange(starts[2] + n48 - n48 % 16, starts[2] + n48), np.arange(0, 16 - n48 % 16)])
    rem = (n48 + n8) % 16
    t2 = np.concatenate([np.arange(n8 - rem, n8), np.arange(starts[1], starts[1] + 16 - rem)])
    res = {"ticks": T, "streams": B, "frames": B * T}
    for gate, (d, dy, near) in results.items():
        res["gate_off" if gate is None else "gate_0.01"] = {
            "kernel_vs_f64_chain": stats(d), "scipy_f32_transform_vs_f64_chain": stats(dy),
            "by_rate": {str(sr): stats(d[starts[k]:starts[k + 1]]) for k, (sr, _, _) in enumerate(rates)},
            "boundary_tiles": {"48k|8k": {"streams": t1.tolist(), **stats(d[t1])}, "8k|24k": {"streams": t2.tolist(), **stats(d[t2])}},
            "resampled_samples_within_2e-6_of_the_gate": near}
    print(json.dumps(res))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutter_vad_amd import weights_io
from cutter_vad_amd.engine import Engine
from oracle import oracle
from tests.signals import make_streams
for sr in (16000, 8000):
    blob = open(weights_io.packaged_blob_path(4, sr), "rb").read()
    om = oracle.OracleModel(blob, "f64")
    for n, T in ((70, 6), (1000, 4)):
        fr = make_streams(n, T, seed=2024 + n)
        for tile in (16, 32):
            with Engine(blob, model_version=4, max_streams=1024, sample_rate=sr) as eng:
                eng.set_tile(tile)
                slots = eng.open_streams(n)
                st = np.zeros((n, 256), np.float32)
                worst = 0.0
                for t in range(T):
                    got = eng.step(slots, np.ascontiguousarray(fr[:, t]), denoise=0.01)
                    ref = om.step_batch(oracle.denoise(fr[:, t]).reshape(n, 512), st, nthreads=8)
                    worst = max(worst, float(np.abs(got - ref).max()))
                hc = np.stack([eng.get_state(int(s)) for s in slots[:8]])
                print(f"V4 {sr} tile{tile} n={n}: max |dp| = {worst:.2e}; state err {np.abs(hc.reshape(8, -1) - st[:8]).max():.2e}", flush=True)

#!/usr/bin/env python3
"""Scratch timing of the V5 step on device-resident frames (B streams, K steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cutter_vad_amd import weights_io
from cutter_vad_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
V = int(sys.argv[3]) if len(sys.argv) > 3 else 5
blob = open(weights_io.packaged_blob_path(V), "rb").read()
eng = Engine(blob, model_version=V, max_streams=B)
eng.open_streams(B)
g = torch.Generator(device="cuda").manual_seed(0)
ring = (0.1 * torch.randn(32, B, 512, device="cuda", generator=g)).contiguous()
probs = torch.empty(B, device="cuda")
ts = torch.cuda.Stream()
torch.cuda.synchronize()
torch.cuda.set_stream(ts)
st = ts.cuda_stream
assert st != 0
for i in range(20):
    eng.step_device(B, ring[i % 32].data_ptr(), probs.data_ptr(), stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(K):
    eng.step_device(B, ring[i % 32].data_ptr(), probs.data_ptr(), stream=st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
fps = B / (ms * 1e-3)
print(f"V{V} B={B} K={K}: {ms*1e3:.1f} us/step  {fps/1e6:.2f} M frames/s  "
      f"{fps*988160/1e12:.1f} TFLOP/s algorithmic ({fps*988160/157.3e12*100:.1f}% of fp32 peak)  probs mean {probs.mean().item():.4f}")

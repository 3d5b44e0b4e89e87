#!/usr/bin/env python3
"""Convert the reference's Silero ``.onnx`` files into the engine's SVW weight blobs.

Usage: python tools/extract_weights.py [MODEL_DIR] [OUT_DIR]
Defaults: /root/reference/src/real_time_vad/models -> cutter_vad_amd/weights

The blobs carry only the float32 tensors of one graph branch (V5 16 kHz / 8 kHz; V4 16 kHz / 8 kHz) under canonical names
(see cutter_vad_amd/weights_io.py); they are data, the graph itself is not copied.
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutter_vad_amd import weights_io  # noqa: E402


def main() -> None:
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/real_time_vad/models"
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cutter_vad_amd", "weights")
    os.makedirs(dst, exist_ok=True)
    for version, fname, sr in ((5, "silero_vad_v5.onnx", 16000), (5, "silero_vad_v5.onnx", 8000),
                               (4, "silero_vad.onnx", 16000), (4, "silero_vad.onnx", 8000)):
        path = os.path.join(src, fname)
        tensors = weights_io.extract_from_onnx(path, version, sr)
        blob = weights_io.pack_svw(version, tensors)
        out = os.path.join(dst, os.path.basename(weights_io.packaged_blob_path(version, sr)))
        with open(out, "wb") as f:
            f.write(blob)
        n = sum(t.size for t in tensors.values())
        src_sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
        print(f"v{version}: {len(tensors)} tensors, {n} params, {len(blob)} B -> {out}")
        print(f"     source sha256 {src_sha}")
        for k in sorted(tensors):
            print(f"     {k:14s} {tuple(tensors[k].shape)}")


if __name__ == "__main__":
    main()

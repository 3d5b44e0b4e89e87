#!/bin/bash
# CPU-only AddressSanitizer + UBSan run of the host-side packers (csrc/pack_weights.cpp): the packaged blobs, truncated
# blobs (must be refused cleanly) and the three resample operators.  GPU sanitizers are not available on the pool.
set -e
OUT=${TMPDIR:-/tmp}/packsan_$$
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Icutter_vad_amd/csrc -o "$OUT" tools/san_packer.cpp cutter_vad_amd/csrc/pack_weights.cpp
"$OUT" cutter_vad_amd/weights/silero_v5_16k.svw cutter_vad_amd/weights/silero_v4_16k.svw cutter_vad_amd/weights/silero_v4_8k.svw
rm -f "$OUT"

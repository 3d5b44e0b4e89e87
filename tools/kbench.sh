#!/bin/bash
# usage: tools/kbench.sh [extra hipcc flags...] -- <B> <steps> [T]      (run from the repo root, on a GPU box)
set -e
FLAGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
[ "$1" == "--" ] && shift
OUT=${KBENCH_OUT:-/tmp/kbench_$$}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=8 "${FLAGS[@]}" -o "$OUT" tools/kbench.cpp cutter_vad_amd/csrc/silero_v5.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
"$OUT" cutter_vad_amd/weights/silero_v5_16k.svw "$@"

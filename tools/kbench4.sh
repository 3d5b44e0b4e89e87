#!/bin/bash
# usage: tools/kbench4.sh [extra hipcc flags...] -- <B> <steps> [8k]      (run from the repo root, on a GPU box)
#        -DKB_TILE16 [-DKB_ONE_PER_CU=1]: the 16-stream tile kernel; -DVADK_STAMPS: per-phase cycle stamps
set -e
FLAGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
[ "$1" == "--" ] && shift
OUT=${KBENCH_OUT:-/tmp/kbench4_$$}
BLOB=cutter_vad_amd/weights/silero_v4_16k.svw
[ "$3" == "8k" ] && BLOB=cutter_vad_amd/weights/silero_v4_8k.svw
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form "${FLAGS[@]}" -o "$OUT" tools/kbench4.cpp cutter_vad_amd/csrc/silero_v4.hip cutter_vad_amd/csrc/silero_v4_t16.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
"$OUT" $BLOB "$1" "$2"

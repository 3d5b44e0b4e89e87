#!/bin/bash
# usage (GPU box, repo root): tools/ubench/run_tile_width.sh   -> 32- vs 16-stream tiles at several VALU loads per block
set -e
for nv in 0 16 48 96; do
  for mode in 32 16; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMODE=$mode -DNV=$nv -o /tmp/tw_$mode tools/ubench/tile_width.hip 2>/dev/null
    /tmp/tw_$mode 1200 | tail -1
  done
done

#!/bin/bash
# run from the repo root on a GPU box: sweeps KIND x NV of tools/ubench/mfma_valu.hip
for k in 0 1 2 3 4 5 6; do
  for nv in 0 2 4 8 12 16; do
    hipcc --offload-arch=gfx950 -O3 -DKIND=$k -DNV=$nv -o /tmp/mv tools/ubench/mfma_valu.hip 2>/dev/null && /tmp/mv || exit 1
  done
done

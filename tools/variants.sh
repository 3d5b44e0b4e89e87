#!/bin/bash
# Same-box A/B of kernel variants through the PRODUCT library (GPU box, repo root):
#   tools/variants.sh <file.hip[,file2.hip]> "<command>" "<defines of variant 1>" "<defines of variant 2>" ...
# For every variant ("" = the product build) the named .hip files are recompiled with the defines, libvad_engine.so is relinked in
# place (the box's copy of the repo is scratch) and <command> runs; twice round, interleaved, because box-to-box and run-to-run
# variation is +-1 us.  The last build is the product build again.
set -e
FILES=$1; CMD=$2; shift 2
for round in 1 2; do
  for v in "$@"; do
    VAD_KERNEL_DEFINES="$v" VAD_KERNEL_DEFINES_FILES="$FILES" python3 cutter_vad_amd/_build.py > /dev/null
    echo "== round $round variant [$v]"
    bash -c "$CMD"
  done
done
VAD_KERNEL_DEFINES="" VAD_KERNEL_DEFINES_FILES="$FILES" python3 cutter_vad_amd/_build.py > /dev/null

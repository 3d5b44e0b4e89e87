#!/bin/bash
# A/B of a model kernel: HEAD against the working tree, on ONE box, interleaved (box-to-box variation is +-1 us, more than
# most changes are worth).
#   here (has .git):   tools/ab_kernel.sh prepare          -> tools/ab_tmp/old/ = csrc of HEAD (untracked; delete afterwards)
#   on the GPU box:    tools/ab_kernel.sh run v4|v5 [B] [steps]
set -e
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form"
if [ "$1" == "prepare" ]; then
  rm -rf tools/ab_tmp && mkdir -p tools/ab_tmp/old
  for f in silero_v4.hip silero_v5.hip vad_layout.h sm_device.h vadk_device.h pack_weights.cpp pack_weights.h; do
    git show HEAD:cutter_vad_amd/csrc/$f > tools/ab_tmp/old/$f
  done
  exit 0
fi
V=$2; B=${3:-8192}; K=${4:-400}
[ "$V" == "v4" ] && { KB=kbench4; SRC=silero_v4.hip; W=silero_v4_16k.svw; } || { KB=kbench; SRC=silero_v5.hip; W=silero_v5_16k.svw; }
sed 's#"../cutter_vad_amd/csrc/#"#' tools/$KB.cpp > tools/ab_tmp/old/$KB.cpp
hipcc $F -Itools/ab_tmp/old -o /tmp/kb_old tools/ab_tmp/old/$KB.cpp tools/ab_tmp/old/$SRC tools/ab_tmp/old/pack_weights.cpp 2>/dev/null
hipcc $F -o /tmp/kb_new tools/$KB.cpp cutter_vad_amd/csrc/$SRC cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
for i in 1 2 3; do
  echo -n "HEAD: "; /tmp/kb_old cutter_vad_amd/weights/$W $B $K | head -1
  echo -n "tree: "; /tmp/kb_new cutter_vad_amd/weights/$W $B $K | head -1
done

#!/bin/bash
# usage (repo root, GPU box): tools/pmc_sq16.sh <tag> [streams=4096]  -> gpurun_out/<tag>_pmc_sq_t16.json
# SQ-side counters and HBM traffic of the 16-stream tile kernel (silero_v5_step16), one rocprofv3 --pmc pass per group
# (--kernel-trace only), over tools/kbench.cpp built with -DKB_TILE16.
set -e
TAG=$1
B=${2:-4096}
OUT=$PWD/gpurun_out
REPO=$PWD
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=8 -DKB_TILE16 -o /tmp/kb16 tools/kbench.cpp cutter_vad_amd/csrc/silero_v5_t16.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
/tmp/kb16 "$REPO/cutter_vad_amd/weights/silero_v5_16k.svw" $B 200 > "$OUT/${TAG}_kb16.log" 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "MfmaUtil" "VALUBusy" "LdsBankConflict" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  KB_RING=32 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/${TAG}_sq16/p$i" -o kb -- /tmp/kb16 "$REPO/cutter_vad_amd/weights/silero_v5_16k.svw" $B 12 > "$OUT/${TAG}_sq16_p$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/${TAG}_sq16_progress.log"
  echo "pass $i done: $grp" >> "$OUT/${TAG}_sq16_progress.log"
done
cd "$REPO"
python3 tools/rocpd_export.py pmc "$OUT/${TAG}_sq16" silero_v5_step16 > "$OUT/${TAG}_pmc_sq_t16.json"

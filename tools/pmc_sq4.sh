#!/bin/bash
# usage (repo root, GPU box): tools/pmc_sq4.sh <tag>  -> gpurun_out/<tag>_pmc_sq_v4.json
# The SQ-side counters of tools/pmc_sq.sh for the V4 kernel (silero_v4_step), over tools/kbench4.
set -e
TAG=$1
OUT=$PWD/gpurun_out
REPO=$PWD
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/kb4 tools/kbench4.cpp cutter_vad_amd/csrc/silero_v4.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "MfmaUtil" "VALUBusy" "LdsBankConflict" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$OUT/${TAG}_sq4/p$i" -o kb -- /tmp/kb4 "$REPO/cutter_vad_amd/weights/silero_v4_16k.svw" 8192 12 > "$OUT/${TAG}_sq4_p$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/${TAG}_sq4_progress.log"
  echo "pass $i done: $grp" >> "$OUT/${TAG}_sq4_progress.log"
done
cd "$REPO"
python3 tools/rocpd_export.py pmc "$OUT/${TAG}_sq4" silero_v4_step > "$OUT/${TAG}_pmc_sq_v4.json"

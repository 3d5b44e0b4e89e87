#!/bin/bash
# usage (repo root, GPU box): tools/pmc_sq4.sh <tag> [t16]  -> gpurun_out/<tag>_pmc_sq_v4[_t16].json
# The SQ-side counters of tools/pmc_sq.sh for the V4 kernels over tools/kbench4: silero_v4_step (32-stream tiles), or with `t16`
# silero_v4_step16 (16-stream tiles, two workgroups per CU).
set -e
TAG=$1
OUT=$PWD/gpurun_out
REPO=$PWD
KFLAGS=""; KNAME=silero_v4_step; SUF=""
if [ "$2" == "t16" ]; then KFLAGS="-DKB_TILE16"; KNAME=silero_v4_step16; SUF="_t16"; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form $KFLAGS -o /tmp/kb4 tools/kbench4.cpp cutter_vad_amd/csrc/silero_v4.hip cutter_vad_amd/csrc/silero_v4_t16.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "MfmaUtil" "VALUBusy" "LdsBankConflict" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$OUT/${TAG}_sq4$SUF/p$i" -o kb -- /tmp/kb4 "$REPO/cutter_vad_amd/weights/silero_v4_16k.svw" 8192 12 > "$OUT/${TAG}_sq4${SUF}_p$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/${TAG}_sq4${SUF}_progress.log"
  echo "pass $i done: $grp" >> "$OUT/${TAG}_sq4${SUF}_progress.log"
done
cd "$REPO"
python3 tools/rocpd_export.py pmc "$OUT/${TAG}_sq4$SUF" $KNAME > "$OUT/${TAG}_pmc_sq_v4$SUF.json"

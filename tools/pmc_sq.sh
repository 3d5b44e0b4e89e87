#!/bin/bash
# usage (repo root, GPU box): tools/pmc_sq.sh <tag>  -> gpurun_out/<tag>_sq/<pass>/..., gpurun_out/<tag>_pmc_sq.json
# SQ-side evidence for the V5 kernel (MFMA utilisation, VALU busy, LDS bank conflicts, instruction counts), one rocprofv3
# --pmc pass per group (the hardware counters do not fit together), --kernel-trace only, over tools/kbench.
set -e
TAG=$1
OUT=$PWD/gpurun_out
REPO=$PWD
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=8 -DKB_TILE16 -o /tmp/kb tools/kbench.cpp cutter_vad_amd/csrc/silero_v5_t16.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "MfmaUtil" "VALUBusy" "LdsBankConflict" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  KB_RING=32 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/${TAG}_sq/p$i" -o kb -- /tmp/kb "$REPO/cutter_vad_amd/weights/silero_v5_16k.svw" 8192 12 > "$OUT/${TAG}_sq_p$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/${TAG}_sq_progress.log"
  echo "pass $i done: $grp" >> "$OUT/${TAG}_sq_progress.log"
done
cd "$REPO"
python3 tools/rocpd_export.py pmc "$OUT/${TAG}_sq" > "$OUT/${TAG}_pmc_sq.json"

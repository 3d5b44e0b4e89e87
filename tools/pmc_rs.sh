#!/bin/bash
# usage (repo root, GPU box): tools/pmc_rs.sh <tag>  -> gpurun_out/<tag>_pmc_sq_rs.json
# SQ-side counters and HBM traffic of the FUSED resample -> step launch (silero_v5_step16<true, true, ...>): 4 096 streams all at
# 48 kHz (tools/bench_configs.py rates48) and configs[3] at its stated size; one rocprofv3 --pmc pass per group, --kernel-trace
# only, the program directly after `--`.
set -e
TAG=$1
OUT=$PWD/gpurun_out
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
export VAD_BENCH_K=24 VAD_BENCH_WU=4
for cfg in rates48 config3; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU" "VALUBusy" "MfmaUtil" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "LdsBankConflict" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace -d "$OUT/${TAG}_rs_${cfg}/p$i" -o rs -- python3 "$REPO/tools/bench_configs.py" $cfg > "$OUT/${TAG}_rs_${cfg}_p$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/${TAG}_rs_progress.log"
    echo "$cfg pass $i done: $grp" >> "$OUT/${TAG}_rs_progress.log"
  done
done
cd "$REPO"
python3 - "$OUT" "$TAG" <<'PY'
import json, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for cfg in ("rates48", "config3"):
    res[cfg] = json.loads(subprocess.check_output([sys.executable, "tools/rocpd_export.py", "pmc", f"{out}/{tag}_rs_{cfg}", "silero_v5_step16<true, true,"]))
json.dump(res, open(f"{out}/{tag}_pmc_sq_rs.json", "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"], 1) for c, v in r["counters"].items()} for k, r in res.items()}, indent=1))
PY

#!/bin/bash
# usage (repo root, GPU box): tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*
# 1. bench.py (full, with the CPU leg)            -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py -> <tag>_bench_kernel_stats.csv
# 3. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) over tools/kbench built for the kernel bench.py runs
#    (silero_v5_step16: one-frame calls are served on 16-stream tiles, two workgroups per CU) -> <tag>_pmc_*.csv
set -e
TAG=$1
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench done" >> "$OUT/${TAG}_progress.log"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=8 -DKB_TILE16 -o /tmp/kb tools/kbench.cpp cutter_vad_amd/csrc/silero_v5_t16.hip cutter_vad_amd/csrc/pack_weights.cpp 2>/dev/null
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_prof" -o bench -- python3 "$REPO/bench.py" --no-cpu --steps 500 --warmup 100 > "$OUT/${TAG}_prof.log" 2>&1
echo "kernel trace done" >> "$OUT/${TAG}_progress.log"
for c in FETCH_SIZE WRITE_SIZE; do
  KB_RING=32 rocprofv3 --pmc $c --kernel-trace -d "$OUT/${TAG}_pmc/$c" -o kb -- /tmp/kb "$REPO/cutter_vad_amd/weights/silero_v5_16k.svw" 8192 20 > "$OUT/${TAG}_pmc_$c.log" 2>&1
  echo "pmc $c done" >> "$OUT/${TAG}_progress.log"
done
cd "$REPO"
# rocprofv3 7.x writes rocpd (sqlite) databases by default: export the summaries profiles/ keeps
python3 tools/rocpd_export.py stats "$(find "$OUT/${TAG}_prof" -name '*_results.db' | head -1)" > "$OUT/${TAG}_bench_kernel_stats.csv"
python3 tools/rocpd_export.py pmc "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc_traffic.json"

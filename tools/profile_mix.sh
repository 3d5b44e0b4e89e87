#!/bin/bash
# usage (repo root, GPU box): tools/profile_mix.sh <tag>  -> gpurun_out/<tag>_mix_*
# configs[4] per-GPU share (bench.py --mix v4v5): the line itself, rocprofv3 --kernel-trace --stats of the same command, and
# the HBM traffic of its two kernels (separate --pmc passes, FETCH_SIZE / WRITE_SIZE, --kernel-trace only; the program goes
# directly after `--`).  With counters on, rocprofv3 serialises the kernels: bytes per launch are what is read off, not times.
set -e
TAG=$1
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p "$OUT"
python3 bench.py --mix v4v5 > "$OUT/${TAG}_bench_mix_v4v5.json" 2> "$OUT/${TAG}_bench_mix.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_mix_prof" -o bench -- python3 "$REPO/bench.py" --mix v4v5 --no-cpu --steps 500 --warmup 100 > "$OUT/${TAG}_mix_prof.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$OUT/${TAG}_mix_pmc/$c" -o kb -- python3 "$REPO/bench.py" --mix v4v5 --no-cpu --steps 40 --warmup 10 > "$OUT/${TAG}_mix_pmc_$c.log" 2>&1
  echo "pmc $c done" >> "$OUT/${TAG}_mix_progress.log"
done
cd "$REPO"
python3 tools/rocpd_export.py stats "$(find "$OUT/${TAG}_mix_prof" -name '*_results.db' | head -1)" > "$OUT/${TAG}_mix_kernel_stats.csv"
python3 - "$OUT" "$TAG" <<'PY'
import json, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for k in ("silero_v5_step", "silero_v4_step16"):
    res[k] = json.loads(subprocess.check_output([sys.executable, "tools/rocpd_export.py", "pmc", f"{out}/{tag}_mix_pmc", k]))
total = sum(r.get("hbm_bytes_per_launch", 0) for r in res.values())
json.dump({"command": "bench.py --mix v4v5 (4 096 Silero V5 streams on 32-stream tiles + 4 096 Silero V4 streams on 16-stream tiles per step)",
           "kernels": res, "hbm_bytes_per_step": total}, open(f"{out}/{tag}_pmc_traffic_mix.json", "w"), indent=1)
PY

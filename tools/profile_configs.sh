#!/bin/bash
# usage (repo root, GPU box): tools/profile_configs.sh <tag>   -> gpurun_out/<tag>_configs_kernel_stats.csv
# rocprofv3 --kernel-trace --stats over the non-headline configs (V4, V4 8 kHz, batch 1 024, mixed-rate resample + V5, resampler alone):
# per-kernel average durations of silero_v4_step16 / vadk_resample_512 / silero_v5_step[16] to set beside tools/bench_configs.py.
set -e
TAG=$1
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_cfgprof" -o cfg -- python3 "$REPO/tools/bench_configs.py" v4_alone v4_8k config1 config3 resampler_alone > "$OUT/${TAG}_cfgprof.log" 2>&1
# the generic whole-array resampler's kernels (direct: vadk_rsg_*; chirp-z / FFT: vadk_rsf_*), in a run of their own
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_rsgprof" -o rsg -- python3 "$REPO/tools/bench_resample_generic.py" > "$OUT/${TAG}_rsgprof.log" 2>&1
cd "$REPO"
python3 tools/rocpd_export.py stats "$(find "$OUT/${TAG}_cfgprof" -name '*_results.db' | head -1)" > "$OUT/${TAG}_configs_kernel_stats.csv"
python3 tools/rocpd_export.py stats "$(find "$OUT/${TAG}_rsgprof" -name '*_results.db' | head -1)" > "$OUT/${TAG}_resample_generic_kernel_stats.csv"

#!/bin/bash
# CPU sanitizer runs of the tick assembler's host logic: csrc/engine.cpp + the packers compiled against the test-only HIP
# stand-in in tools/san_tick/hip/ (GPU sanitizers are not available on the pool).  usage: run.sh [thread|address|plain]...
set -e
cd "$(dirname "$0")/../.."
MODES=${@:-thread address}
OUT=${TMPDIR:-/tmp}/san_tick_$$
SRC="tools/san_tick/san_tick.cpp tools/san_tick/fake_kernels.cpp cutter_vad_amd/csrc/engine.cpp cutter_vad_amd/csrc/pack_weights.cpp cutter_vad_amd/csrc/resample_tables.cpp"
for m in $MODES; do
    case $m in
        thread) F="-fsanitize=thread" ;;
        address) F="-fsanitize=address,undefined -fno-sanitize-recover=undefined" ;;
        *) F="" ;;
    esac
    g++ -std=c++17 -O1 -g -fno-omit-frame-pointer $F -Itools/san_tick -Icutter_vad_amd/csrc -o "$OUT" $SRC -lpthread
    echo "== $m"
    TSAN_OPTIONS="halt_on_error=1" ASAN_OPTIONS="detect_leaks=1" "$OUT" cutter_vad_amd/weights/silero_v5_16k.svw
done
rm -f "$OUT"

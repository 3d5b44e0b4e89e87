"""The serving tick's C side on the CPU: the real engine.cpp (vad_tick_run_work, vad_tick_take_segment_wav16) over stand-in kernels,
the C inbox with its fallback, and the conducting tick (_wirebox.tick_shards) - tests/scripts/conductor_check.py, in its own
process (it loads the stand-in library in place of the HIP one, which must not leak into this process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conducted_tick_equals_the_per_pool_tick_and_builds_wavwriters_bytes():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scripts", "conductor_check.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["same_events"] and r["same_done"], r            # three pools conducted from C == each pool ticked by itself
    assert r["events"] > 300 and r["ends"] >= 30, r          # START / CONTINUE (notifications and payloads) / END all occurred
    assert r["wav_matches_wavwriter"] and r["wav_len"] == r["want_len"] > 44, r
    assert r["forked_child_conducts"], r                     # a forked child starts with an empty crew (no wait for threads it has not got)

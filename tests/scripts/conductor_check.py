"""Run by tests/test_conductor.py in its own process: the serving tick's C side WITHOUT a GPU - the real csrc/engine.cpp (tick
assembler, vad_tick_run_work, vad_tick_take_segment_wav16) over the HIP stand-in of tools/san_tick/ (tests/standin.py: host
stand-ins for the kernel launches, p = |first sample of the frame|) - under the real pools, the C inbox and the conducting tick
(_wirebox.tick_shards).  Prints one JSON object."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cutter_vad_amd import _ffi, VADConfig  # noqa: E402
from tests import standin  # noqa: E402

_ffi.LIB_PATH = standin.build(os.path.join(tempfile.gettempdir(), "vad_standin_lib"))
from cutter_vad_amd.pool import EnginePool  # noqa: E402
from cutter_vad_amd.server import ShardedStreamPool, SharedStreamPool  # noqa: E402
from cutter_vad_amd.server import shared_pool  # noqa: E402
from cutter_vad_amd.utils.wav_writer import WAVWriter  # noqa: E402

assert shared_pool._wirebox is not None, "the C inbox must be built (python -m cutter_vad_amd._build)"


def frame(p, n=480, seed=0):
    x = (np.random.default_rng(seed).standard_normal(n) * 2000).astype("<i2")
    x[0] = int(round(p * 32767))
    return x.tobytes()


def run(conducted: bool):
    shards = [SharedStreamPool(max_streams=64, pool=EnginePool(), device_id=0) for _ in range(3)]
    pool = ShardedStreamPool(shards=shards)
    cfg = VADConfig(vad_start_probability=0.5, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3, buffer_size=480)
    cfg32 = VADConfig(vad_start_probability=0.5, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3, buffer_size=480,
                      output_wav_bit_depth=32)
    log, sess = [], []
    for k in range(30):
        s = pool.open_session(cfg32 if k == 7 else cfg, shard=k % 3)
        payload = (k % 5 == 0)
        s.set_callbacks(lambda k=k: log.append((k, "S")),
                        lambda wav, k=k: log.append((k, "E", len(wav), hashlib.sha256(wav).hexdigest())),
                        (lambda pcm, k=k: log.append((k, "C", len(pcm)))) if k % 2 == 0 else None,
                        continue_payload=payload)
        sess.append(s)
    for t in range(26):
        for k, s in enumerate(sess):
            talking = (t + k) % 13 < 6
            s.submit_pcm16(frame(0.9 if talking else 0.05, seed=1000 * t + k))
            if k == 3 and t == 10:
                s.submit_pcm16(frame(0.9, n=700, seed=5))      # a frame longer than the model's: the general path, in order
        n = pool.tick() if conducted else sum(p._tick_python() for p in pool.shards)
        assert n >= 30
    pool.drain() if conducted else [p.drain() for p in pool.shards]
    done = [s.frames_done for s in sess]
    pool.close()
    return log, done


a_log, a_done = run(True)
b_log, b_done = run(False)
# WAV payloads against WAVWriter on the same segment, taken as float32
eng_pool = SharedStreamPool(max_streams=8, pool=EnginePool(), device_id=0)
s = eng_pool.open_session(VADConfig(vad_start_probability=0.5, vad_end_probability=0.3, voice_start_frame_count=1, voice_end_frame_count=2,
                                    buffer_size=480, enable_denoising=True))
got = []
s.set_callbacks(None, got.append, None)
frames = [frame(0.9, seed=k) for k in range(5)] + [frame(0.0, seed=9)] * 3
for f in frames:
    s.submit_pcm16(f)
    eng_pool.tick()
x = np.concatenate([np.frombuffer(f, "<i2").astype(np.float32) / np.float32(32767.0) for f in frames[:7]])
x = np.where(np.abs(x) > 0.01, x, 0).astype(np.float32)
want = WAVWriter(16000, 16, 1).write_wav_data(x)
eng_pool.close()

# a forked child conducts by itself: it has none of the parent's crew threads (fork copies the calling thread only) and must not wait
# for them (wirebox.c: crew_atfork_child).  The child gets 60 s; a hang is a failure, not a timeout of the whole test.
def forked_child_conducts() -> bool:
    import signal
    pid = os.fork()
    if pid == 0:
        try:
            signal.alarm(60)
            shards = [SharedStreamPool(max_streams=16, pool=EnginePool(), device_id=0) for _ in range(3)]
            pool = ShardedStreamPool(shards=shards)
            ss = [pool.open_session(VADConfig(buffer_size=480), shard=k % 3) for k in range(6)]
            for t in range(3):
                for k, s_ in enumerate(ss):
                    s_.submit_pcm16(frame(0.5, seed=t * 10 + k))
                assert pool.tick() == 6
            os._exit(0)
        except BaseException:
            os._exit(1)
    _, status = os.waitpid(pid, 0)
    return os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


child_ok = forked_child_conducts()
print(json.dumps({"forked_child_conducts": child_ok, "same_events": a_log == b_log, "events": len(a_log), "ends": sum(1 for e in a_log if e[1] == "E"),
                  "same_done": a_done == b_done, "done": a_done[:4], "wav_matches_wavwriter": bool(got and got[0] == want),
                  "wav_len": len(got[0]) if got else 0, "want_len": len(want)}))

"""The shared stream pool on the real engine: every session must behave exactly like its own VADWrapper
(the reference's one-wrapper-per-client design, websocket_service/server/vad_websocket_server.py:277)."""

import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pooled_sessions_equal_private_wrappers():
    from cutter_vad_amd import VADConfig, VADWrapper
    from cutter_vad_amd.server import SharedStreamPool
    pcm = np.load(os.path.join(GOLD, "speech16k_i16.npz"))["pcm"]
    x = pcm.astype(np.float32) / 32767.0
    variants = [dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6, voice_end_frame_count=12, buffer_size=480),
                dict(vad_start_probability=0.5, vad_end_probability=0.35, voice_start_frame_count=3, voice_end_frame_count=8, buffer_size=512),
                dict(vad_start_probability=0.6, vad_end_probability=0.5, voice_start_frame_count=10, voice_end_frame_count=20, buffer_size=480,
                     enable_denoising=False)]
    N = 24
    offs = [(7919 * k) % 16000 for k in range(N)]                # every session hears the clip from another offset
    seconds = 14
    pool = SharedStreamPool()
    logs = [[] for _ in range(N)]
    sessions = []
    try:
        for k in range(N):
            cfg = VADConfig(**variants[k % 3])
            s = pool.open_session(cfg)
            s.set_callbacks(lambda k=k: logs[k].append("S"), lambda wav, k=k: logs[k].append(hashlib.sha256(wav).hexdigest()),
                            lambda b, k=k: logs[k].append(hashlib.sha256(b).hexdigest()[:8]))
            sessions.append(s)
        nfr = [seconds * 16000 // variants[k % 3]["buffer_size"] for k in range(N)]
        probs = [[] for _ in range(N)]
        for i in range(max(nfr)):
            for k, s in enumerate(sessions):
                L = variants[k % 3]["buffer_size"]
                if i < nfr[k]:
                    if k % 2:                                  # half of the clients send int16 wire frames
                        s.submit_pcm16(pcm[offs[k] + i * L: offs[k] + (i + 1) * L].astype("<i2").tobytes())
                    else:
                        s.submit(x[offs[k] + i * L: offs[k] + (i + 1) * L])
            pool.tick()
            for k, s in enumerate(sessions):
                if i < nfr[k]:
                    probs[k].append(s.last_probability)
        assert pool.stats()["frames_per_launch"] > 5         # (wire format, gate) groups get their own launch
        assert sum("S" in lg for lg in logs) == N
        # the same audio through a private wrapper per client
        for k in (0, 1, 2, 5, 13, 23):
            L = variants[k % 3]["buffer_size"]
            ref_log, ref_p = [], []
            with VADWrapper(VADConfig(**variants[k % 3])) as w:
                w.set_callbacks(lambda: ref_log.append("S"), lambda wav: ref_log.append(hashlib.sha256(wav).hexdigest()),
                                lambda b: ref_log.append(hashlib.sha256(b).hexdigest()[:8]))
                for i in range(nfr[k]):
                    w.process_audio_data(x[offs[k] + i * L: offs[k] + (i + 1) * L])
                    ref_p.append(w.processor.voice_probabilities[-1])
            assert probs[k] == ref_p, k          # int16 wire frames included: the kernel's decode is numpy's
            assert logs[k] == ref_log, k
            assert any(len(e) == 64 for e in ref_log), "the scenario must finish at least one segment"
    finally:
        pool.close()

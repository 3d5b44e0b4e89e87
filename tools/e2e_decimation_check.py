#!/usr/bin/env python3
"""Why SURVEY.md §8(c) quotes 65 324 B for the first WAV of the 4-segment known answer and tests/golden pins 64 364 B.

The reference client turns examples/audios/SampleVoiceMono.wav (48 kHz) into 16 kHz with librosa.resample
(websocket_service/clients/python/sources/audio_sources.py:121-133; librosa / soxr are not installed here), the survey used
"a decimation" it does not specify, tools/make_goldens.py takes every third sample.  The reference's own expectation is only
"4 segments" (examples/test_python_vad_client.py:200-223).  This script (container only: reads the reference's WAV; oracle =
TEST INFRASTRUCTURE) runs the V5 oracle + state machine with the client-test configuration over several 48 -> 16 kHz
conversions and prints START / END frames, WAV sizes and the probability of the frame that decides the first segment's length.
"""
import os
import sys
import wave

import numpy as np
import scipy.signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutter_vad_amd import weights_io  # noqa: E402
from oracle import oracle  # noqa: E402

WAV = "/root/reference/examples/audios/SampleVoiceMono.wav"


def run(x16: np.ndarray):
    om = oracle.OracleModel(open(weights_io.packaged_blob_path(5), "rb").read(), "f64")
    sm = oracle.StateMachine(0.4, 0.3, 0.8, 0.95, 6, 12)
    st = np.zeros(256, np.float32)
    ev_log, sizes, probs = [], [], []
    for i in range(x16.size // 480):
        fr = oracle.pad_frame(oracle.denoise(x16[i * 480:(i + 1) * 480]))
        p = om.step(fr, st)
        probs.append(p)
        ev, seg = sm.step(p, 480)
        if ev & 1:
            ev_log.append(("S", i))
        if ev & 2:
            ev_log.append(("E", i))
            sizes.append(44 + 2 * seg)
    return ev_log, sizes, np.array(probs)


def main():
    with wave.open(WAV, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    f = pcm.astype(np.float64) / 32768.0                                   # what soundfile.read returns
    variants = {
        "every 3rd sample, int16 / 32767 (tests/golden)": (pcm[::3].astype(np.float32) / np.float32(32767.0)),
        "every 3rd sample, offset 1": (pcm[1::3].astype(np.float32) / np.float32(32767.0)),
        "every 3rd sample, offset 2": (pcm[2::3].astype(np.float32) / np.float32(32767.0)),
        "mean of 3": f[: f.size // 3 * 3].reshape(-1, 3).mean(1).astype(np.float32),
        "scipy.signal.resample_poly(1, 3)": scipy.signal.resample_poly(f, 1, 3).astype(np.float32),
        "scipy.signal.decimate(3, fir)": scipy.signal.decimate(f, 3, ftype="fir").astype(np.float32),
        "scipy.signal.resample (Fourier)": scipy.signal.resample(f, f.size // 3).astype(np.float32),
    }
    for name, x in variants.items():
        # the client sends int16 PCM: clip * 32767 -> int16, the server divides by 32767 (vad_websocket_server.py:341)
        q = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        x16 = (q.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
        ev, sizes, probs = run(x16)
        print(f"{name}: segments {len(sizes)}  WAV bytes {sizes}  events {ev[:2]}  p[33..35] = {np.round(probs[33:36], 4).tolist()}")


if __name__ == "__main__":
    main()

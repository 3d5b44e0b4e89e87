#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

Needs /root/reference (read-only).  Two sources of truth are used, and the fixture records which:

  * ``onnx_interp``  - oracle/onnx_interp.py executing the reference's own .onnx graphs node by
    node per the ONNX opset-16 spec (float64 evaluation of the fp32 weights).  onnxruntime, which
    the reference calls for this arithmetic, is not installed; see DESIGN.md "Oracle".
  * ``reference``    - the reference's own Python (VADProcessor / VADWrapper / AudioUtils /
    WAVWriter) imported from /root/reference/src.  Its top-level ``import onnxruntime``
    (core/silero_model.py:13) is satisfied by a placeholder module whose ``InferenceSession``
    delegates ``run`` to ``onnx_interp`` — the same seam the reference's tests patch
    (tests/test_silero_model.py:301).  Everything above that seam (framing, gate, padding,
    state machine, callbacks, WAV bytes) is the reference's real code.

Only data is written: inputs that cannot be regenerated from a seed, and expected outputs.
"""

from __future__ import annotations

import hashlib
import json
import os
import sys
import types
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
MODELS = os.path.join(REF, "src/real_time_vad/models")
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.onnx_interp import SileroOnnxSession  # noqa: E402
from tests.signals import gate, make_streams, model_cases, model_cases_8k  # noqa: E402


# --------------------------------------------------------------------------------------
def import_reference():
    stub = types.ModuleType("onnxruntime")

    class SessionOptions:
        pass

    class GraphOptimizationLevel:
        ORT_ENABLE_ALL = 99

    class InferenceSession:
        def __init__(self, path, sess_options=None, providers=None):
            self._s = SileroOnnxSession(path, np.float64)
            self._providers = list(providers or ["CPUExecutionProvider"])

        def get_inputs(self):
            return self._s.get_inputs()

        def get_outputs(self):
            return self._s.get_outputs()

        def get_providers(self):
            return self._providers

        def run(self, names, feeds):
            return self._s.run(names, feeds)

    stub.SessionOptions = SessionOptions
    stub.GraphOptimizationLevel = GraphOptimizationLevel
    stub.InferenceSession = InferenceSession
    stub.get_available_providers = lambda: ["CPUExecutionProvider"]
    sys.modules["onnxruntime"] = stub
    sys.path.insert(0, os.path.join(REF, "src"))
    import real_time_vad  # noqa: F401
    return real_time_vad


def load_speech_16k() -> np.ndarray:
    """examples/audios/SampleVoiceMono.wav (48 kHz int16 mono) -> every third sample (16 kHz int16)."""
    with wave.open(os.path.join(REF, "examples/audios/SampleVoiceMono.wav"), "rb") as w:
        assert w.getframerate() == 48000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return np.ascontiguousarray(pcm[::3])


# --------------------------------------------------------------------------------------
def run_model(version: int, frames: np.ndarray, batched: bool = False, sample_rate: int = 16000):
    """frames [T,512] (one stream) or [B,T,512] (batched=True).  -> probs, final state [.,256].
    ``sample_rate`` is the graph's ``sr`` input: anything but 16000 selects V4's 8 kHz sub-model."""
    path = os.path.join(MODELS, "silero_vad_v5.onnx" if version == 5 else "silero_vad.onnx")
    sess = SileroOnnxSession(path, np.float64)
    if not batched:
        frames = frames[None]
    Bn, T, _ = frames.shape
    sr = np.array([sample_rate], np.int64)
    probs = np.empty((Bn, T), np.float32)
    if version == 5:
        st = np.zeros((2, Bn, 128), np.float32)
        for t in range(T):
            o, st = sess.run(None, {"input": frames[:, t], "state": st, "sr": sr})
            probs[:, t] = o[:, 0]
        final = np.concatenate([st[0], st[1]], axis=1)
    else:
        h = np.zeros((2, Bn, 64), np.float32)
        c = np.zeros((2, Bn, 64), np.float32)
        for t in range(T):
            o, h, c = sess.run(None, {"input": frames[:, t], "h": h, "c": c, "sr": sr})
            probs[:, t] = o[:, 0]
        final = np.concatenate([h[0], h[1], c[0], c[1]], axis=1)
    return (probs, final) if batched else (probs[0], final[0])


def make_model_goldens(speech_i16):
    cases = model_cases(speech_i16)
    for version in (5, 4):
        out = {}
        meta = {}
        for name, c in cases.items():
            p, s = run_model(version, c["frames"])
            out[f"{name}.probs"] = p
            out[f"{name}.state"] = s
            meta[name] = c["regen"]
            print(f"  v{version} {name}: T={len(p)} p[min,max]=({p.min():.4f},{p.max():.4f})")
        # batch: 7 different streams in ONE session.run per frame (free batch dim of the graph)
        fb = gate(make_streams(7, 10, seed=77))
        p, s = run_model(version, fb, batched=True)
        out["batch7.probs"], out["batch7.state"] = p, s
        meta["batch7"] = "gate(tests.signals.make_streams(7,10,seed=77)), one batched run per frame"
        out["_meta"] = np.frombuffer(json.dumps({"source": "onnx_interp float64", "cases": meta}).encode(), np.uint8)
        np.savez_compressed(os.path.join(OUT, f"model_v{version}.npz"), **out)


def make_v4_8k_goldens(speech_i16):
    """V4's else-branch (sr != 16000), fed the same 512-sample frames the reference would feed it."""
    cases = model_cases(speech_i16)
    out, meta = {}, {}
    for name in ("speech_gate", "noise_0.02", "noise_0.3", "harmonic", "zeros", "short400_padded"):
        fr = cases[name]["frames"][:240]
        p, s = run_model(4, fr, sample_rate=8000)
        out[f"{name}.probs"], out[f"{name}.state"] = p, s
        meta[name] = cases[name]["regen"] + " (first 240 frames), sr = 8000"
        print(f"  v4-8k {name}: T={len(p)} p[min,max]=({p.min():.4f},{p.max():.4f})")
    # 24 kHz and 48 kHz take the same branch: identical outputs
    fr = cases["harmonic"]["frames"]
    for sr in (24000, 48000):
        p, s = run_model(4, fr, sample_rate=sr)
        assert np.array_equal(p, out["harmonic.probs"]) and np.array_equal(s, out["harmonic.state"]), sr
    fb = gate(make_streams(7, 10, seed=77))
    p, s = run_model(4, fb, batched=True, sample_rate=8000)
    out["batch7.probs"], out["batch7.state"] = p, s
    meta["batch7"] = "gate(tests.signals.make_streams(7,10,seed=77)), one batched run per frame, sr = 8000"
    out["_meta"] = np.frombuffer(json.dumps({"source": "onnx_interp float64, sr = 8000 (24000 / 48000 verified identical)",
                                             "cases": meta}).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "model_v4_8k.npz"), **out)


def make_v5_8k_goldens(speech_i16):
    """V5's else-branch (sr = 8000) on 256-sample frames - the frame length its graph is built for (with the
    reference's 512-sample frames a 3-D tensor reaches the LSTM and onnxruntime refuses; verified on the interpreter)."""
    path = os.path.join(MODELS, "silero_vad_v5.onnx")
    sr = np.array([8000], np.int64)
    out, meta = {}, {}

    def run(frames):            # [B, T, 256]
        sess = SileroOnnxSession(path, np.float64)
        st = np.zeros((2, frames.shape[0], 128), np.float32)
        p = np.empty(frames.shape[:2], np.float32)
        for t in range(frames.shape[1]):
            o, st = sess.run(None, {"input": frames[:, t], "state": st, "sr": sr})
            p[:, t] = o[:, 0]
        return p, np.concatenate([st[0], st[1]], axis=1)

    for name, fr in model_cases_8k(speech_i16).items():
        p, s = run(fr[None])
        out[f"{name}.probs"], out[f"{name}.state"] = p[0], s[0]
        meta[name] = "tests/signals.py: model_cases_8k"
    fb = gate(make_streams(7, 5, seed=78).reshape(7, 10, 256))
    p, s = run(fb)
    out["batch7.probs"], out["batch7.state"] = p, s
    meta["batch7"] = "gate(tests.signals.make_streams(7,5,seed=78).reshape(7,10,256)), one batched run per frame"
    try:
        SileroOnnxSession(path, np.float64).run(None, {"input": np.zeros((1, 512), np.float32),
                                                       "state": np.zeros((2, 1, 128), np.float32), "sr": sr})
        raise SystemExit("V5 at sr = 8000 accepted a 512-sample frame: SURVEY a9 says it cannot")
    except ValueError:
        pass
    out["_meta"] = np.frombuffer(json.dumps({"source": "onnx_interp float64, silero_vad_v5.onnx, sr = 8000, 256-sample frames",
                                             "cases": meta}).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "model_v5_8k.npz"), **out)
    print(f"  model_v5_8k: {sorted(k for k in out if k.endswith('.probs'))}")


# --------------------------------------------------------------------------------------
def make_state_machine_goldens(rtv):
    from real_time_vad.core.config import VADConfig, SampleRate, SileroModelVersion
    from real_time_vad.core.silero_model import VADProcessor

    def run(cfg_kwargs, probs, frame_len=512):
        cfg = VADConfig(sample_rate=SampleRate.SAMPLERATE_16, model_version=SileroModelVersion.V5,
                        enable_denoising=False, buffer_size=max(256, frame_len), **cfg_kwargs)
        proc = VADProcessor(cfg)
        it = iter(probs)
        object.__setattr__(proc.model, "predict", lambda frame, sr: float(next(it)))
        events, wavs = [], []
        frame = np.full(frame_len, 0.25, np.float32)
        for _ in probs:
            r = proc.process_frame(frame)
            ev = (1 if r.voice_started else 0) | (2 if r.voice_ended else 0) | (4 if r.voice_continuing else 0)
            events.append(ev)
            wavs.append(len(r.wav_data) if r.wav_data else 0)
        return events, wavs

    rng = np.random.default_rng(2025)
    scen = {}

    def add(name, cfg, probs, frame_len=512):
        ev, wv = run(cfg, probs, frame_len)
        scen[name] = dict(config=cfg, frame_len=frame_len, probs=[float(p) for p in probs], events=ev, wav_bytes=wv)
        print(f"  sm {name}: {len(probs)} frames, starts={sum(1 for e in ev if e & 1)} ends={sum(1 for e in ev if e & 2)}")

    # the reference's own integration scenario (tests/test_silero_model.py:870-976): 3/5 frame counts
    cfg35 = dict(vad_start_probability=0.5, vad_end_probability=0.5, voice_start_frame_count=3, voice_end_frame_count=5)
    add("silence_voice_silence", cfg35, [0.1] * 5 + [0.9] * 10 + [0.1] * 12)
    add("defaults_long", {}, [0.05] * 4 + [0.95] * 30 + [0.2] * 70 + [0.9] * 15 + [0.1] * 60)
    add("flicker", cfg35, [0.9, 0.9, 0.1, 0.9, 0.9, 0.9, 0.6, 0.4, 0.6, 0.4, 0.4, 0.4, 0.4, 0.4, 0.9, 0.9, 0.9, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1])
    add("threshold_equalities", dict(vad_start_probability=0.7, vad_end_probability=0.7, voice_start_frame_count=2, voice_end_frame_count=2),
        [0.7, 0.7, 0.7, 0.6999999, 0.7, 0.69, 0.69, 0.7, 0.7, 0.1, 0.1])
    add("start_count_over_20_never_fires", dict(voice_start_frame_count=21, voice_end_frame_count=5), [0.99] * 60)
    add("end_count_over_100_never_fires", dict(voice_start_frame_count=2, voice_end_frame_count=101), [0.99] * 3 + [0.01] * 150)
    add("end_count_100_fires", dict(voice_start_frame_count=2, voice_end_frame_count=100), [0.99] * 3 + [0.01] * 130)
    add("client_config_480", dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6, voice_end_frame_count=12),
        list(rng.uniform(0, 1, 400)), frame_len=480)
    for i in range(6):
        cfg = dict(vad_start_probability=float(rng.uniform(0.2, 0.8)), vad_end_probability=float(rng.uniform(0.2, 0.8)),
                   voice_start_frame_count=int(rng.integers(1, 12)), voice_end_frame_count=int(rng.integers(1, 30)),
                   voice_start_ratio=float(rng.uniform(0.5, 1.0)), voice_end_ratio=float(rng.uniform(0.5, 1.0)))
        # smoothed random walk so that runs of consecutive frames occur
        z = np.cumsum(rng.standard_normal(600) * 0.35)
        p = 1 / (1 + np.exp(-(z - z.mean())))
        add(f"random_{i}", cfg, list(p.astype(np.float32)))
    with open(os.path.join(OUT, "state_machine.json"), "w") as f:
        json.dump({"source": "reference VADProcessor (core/silero_model.py:790-949), predict() scripted", "scenarios": scen}, f)


# --------------------------------------------------------------------------------------
def make_e2e_golden(rtv, speech_i16):
    """The reference's only real-model expectation: 4 segments on SampleVoiceMono.wav with the
    client-test config (examples/test_python_vad_client.py:89-103, 200-223), server framing
    (websocket_service/server/vad_websocket_server.py:261-274, 341)."""
    from real_time_vad import VADWrapper, VADConfig, SampleRate, SileroModelVersion
    cfg = VADConfig(sample_rate=SampleRate(16000), model_version=SileroModelVersion.V5, vad_start_probability=0.4,
                    vad_end_probability=0.3, voice_start_ratio=0.8, voice_end_ratio=0.95, voice_start_frame_count=6,
                    voice_end_frame_count=12, enable_denoising=True, auto_convert_sample_rate=True, buffer_size=480)
    vad = VADWrapper(config=cfg)
    log = []
    wavs = []
    vad.set_callbacks(voice_start_callback=lambda: log.append(("start", frame_no[0])),
                      voice_end_callback=lambda w: (log.append(("end", frame_no[0])), wavs.append(w)),
                      voice_continue_callback=lambda b: None)
    frame_no = [0]
    probs = []
    nchunks = speech_i16.size // 480
    for i in range(nchunks):
        frame_no[0] = i
        chunk = speech_i16[i * 480:(i + 1) * 480].astype(np.float32) / 32767.0
        vad.process_audio_data(chunk)
        probs.append(vad.processor.voice_probabilities[-1])
    print(f"  e2e: {nchunks} chunks, events={log}, wav sizes={[len(w) for w in wavs]}")
    np.savez_compressed(
        os.path.join(OUT, "e2e_client_config.npz"),
        probs=np.asarray(probs, np.float32),
        event_frames=np.asarray([f for _, f in log], np.int32),
        event_kinds=np.asarray([1 if k == "start" else 2 for k, _ in log], np.int32),
        wav_sizes=np.asarray([len(w) for w in wavs], np.int64),
        wav_sha256=np.asarray([hashlib.sha256(w).hexdigest() for w in wavs]),
        _meta=np.frombuffer(json.dumps({
            "source": "reference VADWrapper/VADProcessor/AudioUtils/WAVWriter; session.run -> onnx_interp float64",
            "config": "16 kHz, V5, buffer_size 480, start 0.4 / end 0.3, 6 / 12 frames, ratios 0.8 / 0.95, denoise on",
            "input": "speech16k_i16.npz in 480-sample chunks, float32 = int16 / 32767.0"}).encode(), np.uint8))
    return wavs


# --------------------------------------------------------------------------------------
def make_util_goldens(rtv):
    from real_time_vad.utils.audio import AudioUtils
    from real_time_vad.utils.wav_writer import WAVWriter
    out = {}
    # resampler (a11): the reference function itself (-> scipy.signal.resample)
    for sr, n_in in ((8000, 256), (24000, 768), (48000, 1536)):
        x = (0.5 * np.random.default_rng(sr).standard_normal((5, n_in))).astype(np.float32)
        y = np.stack([AudioUtils.resample_audio(r, sr, 16000) for r in x])
        assert y.shape == (5, 512) and y.dtype == np.float32
        out[f"resample_{sr}"] = y
    # whole-array resample of odd lengths (edge cases of the Nyquist handling)
    for n_in, n_out in ((1000, 333), (333, 1000), (480, 160), (161, 483)):
        x = (0.5 * np.random.default_rng(n_in).standard_normal(n_in)).astype(np.float32)
        import scipy.signal
        out[f"resample_len_{n_in}_{n_out}"] = scipy.signal.resample(x, n_out).astype(np.float32)
    # framing (a2)
    fr = {}
    for n in (512, 513, 767, 768, 1024, 1500, 4096, 256, 300, 511):
        f = AudioUtils.split_into_frames(np.arange(n, dtype=np.float32), 512, 256)
        fr[str(n)] = [int(f.shape[0])] + [int(r[0]) for r in f]
    f480 = AudioUtils.split_into_frames(np.arange(480, dtype=np.float32), 480, 240)
    fr["480@480/240"] = [int(f480.shape[0])] + [int(r[0]) for r in f480]
    out["_framing"] = np.frombuffer(json.dumps(fr).encode(), np.uint8)
    # gate (a4)
    g = np.array([0.0, 0.01, -0.01, 0.010000001, -0.0100001, 0.5, -0.5, 0.0099999, 1e-9], np.float32)
    out["gate_in"], out["gate_out"] = g, AudioUtils.denoise_audio(g).astype(np.float32)
    # WAV payload (f1)
    x = np.clip(0.6 * np.random.default_rng(9).standard_normal(2000), -1.3, 1.3).astype(np.float32)
    w = WAVWriter(16000, 16, 1).write_wav_data(x)
    out["wav_sha256"] = np.asarray(hashlib.sha256(w).hexdigest())
    out["wav_len"] = np.asarray(len(w))
    out["wav_header"] = np.frombuffer(w[:44], np.uint8)
    np.savez_compressed(os.path.join(OUT, "utils.npz"), **out)
    print("  utils: resample/framing/gate/wav written")


def resample_generic_cases():
    """(name, input array, original_rate, target_rate) of tests/golden/resample_generic.npz; inputs come from seeds, so
    the fixture holds the reference's outputs only.  Shared with tests/signals.py users through this one definition."""
    from tests.signals import resample_generic_inputs
    return resample_generic_inputs()


def make_resample_generic_goldens(rtv):
    """AudioUtils.resample_audio (the reference's function, audio.py:19-55) on whole arrays of arbitrary length and rate."""
    import warnings
    from real_time_vad.utils.audio import AudioUtils
    out = {}
    for name, x, r0, r1 in resample_generic_cases():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = AudioUtils.resample_audio(x, r0, r1)
        assert y.dtype == np.float32
        out[name] = y
    np.savez_compressed(os.path.join(OUT, "resample_generic.npz"), **out)
    print(f"  resample_generic: {len(out)} cases written")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-resample-generic" in sys.argv:
        make_resample_generic_goldens(import_reference())
        return
    speech = load_speech_16k()
    if "--only-v4-8k" in sys.argv:
        make_v4_8k_goldens(speech)
        return
    if "--only-v5-8k" in sys.argv:
        make_v5_8k_goldens(speech)
        return
    np.savez_compressed(os.path.join(OUT, "speech16k_i16.npz"), pcm=speech,
                        _meta=np.frombuffer(b"examples/audios/SampleVoiceMono.wav [::3] (48 kHz -> 16 kHz), int16", np.uint8))
    print(f"speech: {speech.size} samples @16k")
    rtv = import_reference()
    print("model goldens (onnx_interp):")
    make_model_goldens(speech)
    make_v4_8k_goldens(speech)
    make_v5_8k_goldens(speech)
    print("state machine goldens (reference VADProcessor):")
    make_state_machine_goldens(rtv)
    print("end-to-end golden (reference VADWrapper over onnx_interp):")
    make_e2e_golden(rtv, speech)
    make_util_goldens(rtv)
    make_resample_generic_goldens(rtv)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} B")


if __name__ == "__main__":
    main()

"""The only per-frame numbers in the reference that a real onnxruntime produced: the twelve probabilities its README prints
for `examples/probability_demo.py` (/root/reference/README.md:258-273).  The demo's noise is UNSEEDED
(/root/reference/examples/probability_demo.py:57-68: `np.random.randn` on a deterministic three-harmonic "voice"), so the pin
is statistical: replay the demo's protocol for many seeds and require every README value to lie inside the min - max envelope
of the replays.  It is coarse, and it is the one onnxruntime-produced pin there is.

Protocol (probability_demo.py:45-86): ONE Silero V5 16 kHz stream, zero initial state, patterns in order
("Silence", 0.005, 3 chunks), ("Soft voice", 0.3, 4), ("Loud voice", 0.7, 5), ("Background", 0.02, 2) - 14 chunks of 512 samples,
"Frames processed: 14" in the README; the README shows the first twelve.  A voice chunk is
(0.4 sin 150 Hz + 0.3 sin 300 Hz + 0.2 sin 600 Hz + 0.1 randn) * amplitude with t restarting at 0, the others randn * amplitude.
Per chunk the demo calls `vad.processor.process_frame(audio)` (prints its probability) and then `vad.process_audio_data(audio)`
- the same audio through the same LSTM again.  What the README's run did per printed frame is not recorded (the demo as
committed cannot print at all: `.get()` on a pydantic model), so four READINGS are replayed: denoise gate on / off
(`VADConfig.enable_denoising` defaults to True, config.py:97-100) x one / two model steps per printed frame (two = the printed
step followed by a second step on the same audio).

TEST INFRASTRUCTURE (data + recipe); the model itself comes from the caller (oracle on the CPU, the HIP engine on the GPU).
"""
from __future__ import annotations

import numpy as np

# /root/reference/README.md:258-273, in print order (3 silence, 4 soft voice, 5 loud voice); printed with "%.3f"
README_PROBS = (0.187, 0.105, 0.046, 0.928, 0.996, 0.998, 0.999, 0.994, 0.995, 0.996, 0.994, 0.992)
PRINT_HALF_ULP = 0.0005           # a printed 0.999 is any value in [0.9985, 0.9995]
PATTERNS = (("Silence", 0.005, 3), ("Soft voice", 0.3, 4), ("Loud voice", 0.7, 5), ("Background", 0.02, 2))
READINGS = (("gate_off_1step", False, 1), ("gate_off_2steps", False, 2), ("gate_on_1step", True, 1), ("gate_on_2steps", True, 2))
# readings under which every README value must lie inside the envelope; under the fourth (gate on, one step per printed frame)
# one value is known to fall outside - soft-voice frame 4, printed 0.999 - which the tests record rather than hide
CONSISTENT = ("gate_off_1step", "gate_off_2steps", "gate_on_2steps")
N_SEEDS = 300


def demo_chunks(seed: int) -> np.ndarray:
    """[14, 512] float32: the demo's chunks for one seeding of its noise (same draw order as the script)."""
    rng = np.random.RandomState(seed)
    t = np.arange(512) / 16000.0
    out = []
    for name, amplitude, chunks in PATTERNS:
        for _ in range(chunks):
            if "voice" in name.lower():
                signal = np.sin(2 * np.pi * 150 * t) * 0.4 + np.sin(2 * np.pi * 300 * t) * 0.3 + np.sin(2 * np.pi * 600 * t) * 0.2
                noise = rng.randn(512) * 0.1
                audio = (signal + noise) * amplitude
            else:
                audio = rng.randn(512) * amplitude
            out.append(audio.astype(np.float32))
    return np.stack(out)


def all_chunks(n_seeds: int = N_SEEDS) -> np.ndarray:
    """[n_seeds, 14, 512]"""
    return np.stack([demo_chunks(s) for s in range(n_seeds)])


def replay(step, chunks: np.ndarray, steps_per_frame: int) -> np.ndarray:
    """`step(frames[n, 512]) -> probs[n]` advances the n independent streams by one frame.  Returns the PRINTED probabilities
    [n, 14]: the first step on each chunk; with steps_per_frame == 2 the chunk is then stepped once more, unprinted."""
    n, T, _ = chunks.shape
    printed = np.empty((n, T), np.float32)
    for t in range(T):
        fr = np.ascontiguousarray(chunks[:, t])
        printed[:, t] = step(fr)
        for _ in range(steps_per_frame - 1):
            step(fr)
    return printed


def envelope(printed: np.ndarray) -> dict:
    lo, hi = printed.min(axis=0), printed.max(axis=0)
    k = len(README_PROBS)
    readme = np.array(README_PROBS)
    inside = (readme >= lo[:k] - PRINT_HALF_ULP) & (readme <= hi[:k] + PRINT_HALF_ULP)
    return {"min": [float(v) for v in lo], "max": [float(v) for v in hi], "median": [float(v) for v in np.median(printed, axis=0)],
            "readme": list(README_PROBS), "readme_inside": [bool(b) for b in inside],
            "misses": [int(i) + 1 for i in np.nonzero(~inside)[0]]}

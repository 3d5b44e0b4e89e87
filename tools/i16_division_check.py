#!/usr/bin/env python3
"""Exhaustive check of vadk_device.h's i16_div: for every int16 s and d in {32767, 32768},
    q = s * r;  e = fma(-q, d, s);  q' = fma(e, r, q)        (r = float32(1 / d))
equals the IEEE float32 quotient float32(s) / float32(d) that numpy's true division computes (the reference server's
`np.int16 -> float32 / 32767.0`, vad_websocket_server.py:341).  float32 multiply and fma are emulated exactly with rationals.

    python3 tools/i16_division_check.py          -> prints the mismatch counts (0 / 0; a plain multiply by r misses 1 536 values)
"""
from fractions import Fraction

import numpy as np


def round_f32(fr: Fraction) -> np.float32:
    """round an exact rational to the nearest float32, ties to even"""
    if fr == 0:
        return np.float32(0)
    y = np.float32(float(fr))
    cands = [y, np.nextafter(y, np.float32(np.inf)), np.nextafter(y, np.float32(-np.inf))]
    return np.float32(min(cands, key=lambda v: (abs(Fraction(float(v)) - fr), int(np.float32(v).view(np.uint32)) & 1)))


def mismatches(d: float):
    r = np.float32(1.0) / np.float32(d)
    R, D = Fraction(float(r)), Fraction(d)
    bad = plain = 0
    for s in range(-32768, 32768):
        true = np.float32(s) / np.float32(d)
        q = round_f32(Fraction(s) * R)
        e = round_f32(Fraction(s) - Fraction(float(q)) * D)
        q2 = round_f32(Fraction(float(q)) + Fraction(float(e)) * R)
        bad += int(q2 != true)
        plain += int(q != true)
    return bad, plain


if __name__ == "__main__":
    for d in (32767.0, 32768.0):
        bad, plain = mismatches(d)
        print(f"d = {d:.0f}: corrected {bad} mismatches, plain multiply {plain}")

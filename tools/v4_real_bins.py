#!/usr/bin/env python3
"""Which stage of Silero V4's front end loses the bits?  (CPU only; evidence for DESIGN.md §3 "Numerics".)

V4 feeds log(1 + |X| * 2^20) of the 8-column STFT to its first layer.  This script evaluates the STFT of the
bench's synthetic streams three ways - float64 with the stored basis (the oracle's arithmetic), the HIP kernel's
order in float32 (window multiply, 4-way fold, K = 64 accumulation: an emulation, fused multiply-adds rounded once),
and that same float32 evaluation with ONLY the two real-valued bins k = 0 and k = 128 replaced by their exact sums -
and reports the worst deviation of the log-spectrum per frame.

    python3 tools/v4_real_bins.py [streams=2048] [frames=12]

Finding (2 048 x 12 frames): all bins in float32: max |dlog| 0.108 (stream 1641, frame 3, column 4, bin 0:
|X0| = 2.6e-6 from terms of size 0.1), 58 frames > 1e-3;  bins 0 and 128 exact: max 4.1e-4, none > 1e-3.
A real bin cancels to within d of zero with probability ~ d, a complex bin with probability ~ d^2: the ill-conditioned
inputs of the graph are its two real bins, which is why silero_v4.hip sums exactly those in float64.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutter_vad_amd import weights_io  # noqa: E402
from tests.signals import gate, make_streams  # noqa: E402

f32 = np.float32


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    TT = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    _, T = weights_io.unpack_svw(open(weights_io.packaged_blob_path(4), "rb").read())
    B = T["stft.basis"]
    wf = B[0]
    nn = np.arange(1, 64)
    kk = np.arange(129)
    ct = np.cos(2 * np.pi * kk[:, None] * nn[None, :] / 256).astype(f32)
    st = (-np.sin(2 * np.pi * kk[:, None] * nn[None, :] / 256)).astype(f32)
    ev = kk % 2 == 0
    sg_e = np.where((kk // 2) % 2 == 0, 1.0, -1.0).astype(f32)
    sg_o = np.where(((kk - 1) // 2) % 2 == 0, 1.0, -1.0).astype(f32)

    def cols_of(X):
        xp = np.concatenate([X[:, 96:0:-1], X, X[:, 510:414:-1]], axis=1)
        return np.stack([xp[:, 64 * t:64 * t + 256] for t in range(8)], axis=1)

    def fold_f32(c):
        y = (c * wf).astype(f32)
        y1, y2, y3, y4 = y[..., nn], y[..., 128 - nn], y[..., 128 + nn], y[..., 256 - nn]
        s14, d14, s23, d23 = y1 + y4, y1 - y4, y2 + y3, y2 - y3
        pe, po, qe, qo = s14 + s23, s14 - s23, d14 - d23, d14 + d23
        y128, a64, b64 = y[..., 128], y[..., 64] + y[..., 192], y[..., 64] - y[..., 192]
        re = np.where(ev, y128[..., None] + sg_e * a64[..., None], -y128[..., None]).astype(f32)
        im = np.where(ev, 0, -sg_o * b64[..., None]).astype(f32)
        for j in range(63):
            P = np.where(ev, pe[..., j:j + 1], po[..., j:j + 1])
            Q = np.where(ev, qe[..., j:j + 1], qo[..., j:j + 1])
            re = (re.astype(np.float64) + P.astype(np.float64) * ct[:, j]).astype(f32)
            im = (im.astype(np.float64) + Q.astype(np.float64) * st[:, j]).astype(f32)
        alt = np.where(nn % 2 == 0, 1.0, -1.0)
        re[..., 128] = (pe.astype(np.float64) * alt).sum(-1).astype(f32) + y128 + a64
        im[..., 128] = 0
        return re, im

    def lg(m):
        return np.log(1 + m * 2 ** 20)

    fr = make_streams(S, TT, seed=4242)
    tot_a, tot_b = [], []
    for s0 in range(0, S, 64):
        X = gate(fr[s0:s0 + 64].reshape(-1, 512))
        c = cols_of(X)
        R = c.astype(np.float64) @ B.astype(np.float64).T
        m64 = np.sqrt(R[..., :129] ** 2 + R[..., 129:] ** 2)
        re, im = fold_f32(c)
        mf = np.sqrt(re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2)
        dl = np.abs(lg(mf) - lg(m64))
        dlb = dl.copy()
        dlb[..., 0] = 0
        dlb[..., 128] = 0
        tot_a.append(dl.reshape(len(X), -1).max(1))
        tot_b.append(dlb.reshape(len(X), -1).max(1))
    for name, v in (("float32, all bins", np.concatenate(tot_a)), ("float32, bins 0 and 128 exact", np.concatenate(tot_b))):
        print(f"{name}: frames {len(v)}  max |dlog| {v.max():.3e}  p99.99 {np.percentile(v, 99.99):.3e}  "
              f"> 1e-2: {(v > 1e-2).sum()}  > 1e-3: {(v > 1e-3).sum()}  > 1e-4: {(v > 1e-4).sum()}")


if __name__ == "__main__":
    main()

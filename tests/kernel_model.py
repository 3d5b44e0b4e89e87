"""NumPy model of the fused V5 kernel's *dataflow* (test infrastructure, CPU only).

It consumes the packed per-wave weight streams exactly as ``silero_v5.hip`` does — same
section offsets, same block order, same quad/row addressing, same MFMA fragment convention —
but evaluates every v_mfma_f32_32x32x2_f32 group as a float64 tensor contraction.  If the host
packer (csrc/pack_weights.cpp) and the kernel's indexing disagree anywhere, this model
disagrees with the oracle; it lets the CPU-only suite validate the layout without a GPU.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from cutter_vad_amd import _ffi

S_STFT, S_NYQ, S_ENC0, S_ENC1, S_ENC2, S_ENC3, S_LSTM, S_HEADB = range(8)
MT = 32


def packed_streams(version: int, blob: bytes):
    """-> (W [blocks, 64, 4], section offsets [4 waves, 16]) exactly as vad_engine_create uploads them"""
    lib = _ffi.lib()
    n = C.c_size_t()
    sect = (C.c_uint32 * 64)()
    rc = lib.vad_debug_pack_weights(version, blob, len(blob), None, 0, C.byref(n), sect)
    if rc != 0:
        raise RuntimeError(lib.vad_last_create_error().decode())
    out = np.empty(n.value, np.float32)
    rc = lib.vad_debug_pack_weights(version, blob, len(blob), out.ctypes.data_as(C.POINTER(C.c_float)), out.size,
                                    C.byref(n), sect)
    assert rc == 0
    return out.reshape(-1, 64, 4), np.array(sect, dtype=np.int64).reshape(4, 16)


def _mfma4(wblk, a):
    """wblk [64,4] (lane=(h*32+n')), a [64,4] (lane=(h*32+m)) -> D[n',m] contribution."""
    w = wblk.astype(np.float64).reshape(2, 32, 4)
    x = a.astype(np.float64).reshape(2, 32, 4)
    return np.einsum("hni,hmi->nm", w, x)


def _rows(region, row_lo, row_hi):
    """activation fragment of one k-iteration: lanes h=0 read row_lo, lanes h=1 read row_hi."""
    return np.concatenate([region[row_lo], region[row_hi]], axis=0)  # [64,4]


def _vec(blocks):
    """4 lane-expanded vector blocks -> per-channel vector [32] (channel = 8g + 4h + i)."""
    v = np.zeros(32)
    for g in range(4):
        for h in range(2):
            v[8 * g + 4 * h:8 * g + 4 * h + 4] = blocks[g][h * 32]
    return v


def _store_tile(region, row0, acc, relu=True):
    v = np.maximum(acc, 0) if relu else acc
    for g in range(4):
        for h in range(2):
            region[row0 + 2 * g + h] = v[8 * g + 4 * h:8 * g + 4 * h + 4].T  # [32 m, 4]


def v5_step(W, sect, x, hc, gate=0.01, k8=False):
    """x [32,512] f32, hc [32,256] -> (prob [32], new hc [32,256]).  float64 contractions.
    Mirrors silero_v5.hip: folded loader, one activation region RX (row map in vad_layout.h).
    k8: the 8 kHz sub-model instantiation (x [32,256]; window 128, hop 64, waves 0 / 1 own the 64 complex bins)."""
    N = 128 if k8 else 256                      # window
    H, Q4 = N // 2, N // 4
    QL = Q4 // 4                                # quad rows per folded operand (16 | 8)
    CS, PS, ROWN = 4 * QL, (16 if k8 else 32), (80 if k8 else 160)
    NJ, NJ0 = (4, 8) if k8 else (8, 16)
    x = x.astype(np.float64)
    if gate is not None and gate >= 0:
        x = np.where(np.abs(x) > gate, x, 0.0)
    RX = np.zeros((260, 32, 4))
    RH = np.zeros((32, 32, 4))
    RE = RX[164:]
    # loader: window + 4-way fold of every column (vad_layout.h, v5): rows 64c + {0,16,32,48} + q hold pe, po, qe, qo
    wtab = W[sect[0][S_NYQ]].reshape(-1)[:N].astype(np.float64)           # w[n], the k = 0 row of the stored basis
    fcor = np.zeros((3, 3, 32))
    for c in range(3):
        y = x[:, H * c:H * c + N] * wtab[None, :]
        n = np.arange(Q4)
        y1, y2, y3, y4 = y[:, n], y[:, H - n], y[:, H + n], y[:, (N - n) % N]
        pe, po = y1 + y4 + y2 + y3, y1 + y4 - y2 - y3
        qe, qo = y1 - y4 - y2 + y3, y1 - y4 + y2 - y3
        for arr in (pe, po, qe, qo):
            arr[:, 0] = 0.0                                                   # n = 0 is not part of the folded sums
        if k8:
            for k, arr in enumerate((pe, po, qe, qo)):
                RX[CS * c + QL * k:CS * c + QL * k + QL] = arr.reshape(32, QL, 4).transpose(1, 0, 2)
        else:
            # 16 kHz: po | qo as they are (rows 0..15 | 16..31); the even bins' operands fold once more about n = 32 (8 rows each):
            # pe+ 32.., pe- 40.., qe- 48.., qe+ 56..; slot 0 carries the unpaired n = 32 (pe[32] in pe+, qe[32] in qe+)
            mm = np.arange(32)
            pep, pen = pe[:, mm] + pe[:, (64 - mm) % 64], pe[:, mm] - pe[:, (64 - mm) % 64]
            qen, qep = qe[:, mm] - qe[:, (64 - mm) % 64], qe[:, mm] + qe[:, (64 - mm) % 64]
            pep[:, 0], pen[:, 0], qen[:, 0], qep[:, 0] = pe[:, 32], 0.0, 0.0, qe[:, 32]
            RX[CS * c:CS * c + 16] = po.reshape(32, 16, 4).transpose(1, 0, 2)
            RX[CS * c + 16:CS * c + 32] = qo.reshape(32, 16, 4).transpose(1, 0, 2)
            for k, arr in enumerate((pep, pen, qen, qep)):
                RX[CS * c + 32 + 8 * k:CS * c + 40 + 8 * k] = arr.reshape(32, 8, 4).transpose(1, 0, 2)
        fcor[c, 0], fcor[c, 1], fcor[c, 2] = y[:, H], y[:, Q4] + y[:, H + Q4], y[:, Q4] - y[:, H + Q4]
    RH[:] = hc[:, :128].astype(np.float64).reshape(32, 32, 4).transpose(1, 0, 2)
    c_prev = hc[:, 128:].astype(np.float64)
    # bin 128: alternating sum of pe (16 kHz: of the pe+ rows, whose slot 0 is pe[32]) + rank-1 terms
    nyq = np.zeros((3, 32))
    for c in range(3):
        pe = RX[CS * c:CS * c + QL] if k8 else RX[CS * c + 32:CS * c + 40]
        alt = (pe[:, :, 0] - pe[:, :, 1] + pe[:, :, 2] - pe[:, :, 3]).sum(0)
        nyq[c] = np.abs(alt + fcor[c, 0] + fcor[c, 1])
    # STFT: wave w owns bins bin_of_channel(32w + r); even bins contract pe / qe, odd bins po / qo
    mags = {}
    sgn = np.where(np.arange(32) % 2 == 0, 1.0, -1.0)[:, None]                # (-1)^r per tile row
    if k8:
        # 64 complex bins = four 16-row tiles on v_mfma_f32_16x16x4_f32, one per wave (pack_dft4_wave_128_t16): wave w owns the
        # channels 16 w + r (waves 0 / 1 even bins: pe / qe, 2 / 3 odd bins: po / qo); two k-iterations of 16 folded samples
        sgn16 = np.where(np.arange(16) % 2 == 0, 1.0, -1.0)[:, None]
        mg8 = {}
        for w in range(4):
            ws = sect[w][S_STFT]
            even = w < 2
            rR, rI = (0, 2 * QL) if even else (QL, 3 * QL)
            mg = []
            for c in range(3):
                are, aim = np.zeros((16, 32)), np.zeros((16, 32))
                for j in range(2):
                    wre = W[ws + 2 * j].astype(np.float64).reshape(4, 16, 4)        # [kq][row][i]
                    wim = W[ws + 2 * j + 1].astype(np.float64).reshape(4, 16, 4)
                    u = RX[CS * c + rR + 4 * j:CS * c + rR + 4 * j + 4]              # [kq][stream][i]
                    v = RX[CS * c + rI + 4 * j:CS * c + rI + 4 * j + 4]
                    are += np.einsum("kri,kmi->rm", wre, u)
                    aim += np.einsum("kri,kmi->rm", wim, v)
                y128, a64, b64 = fcor[c, 0][None, :], fcor[c, 1][None, :], fcor[c, 2][None, :]
                if even:
                    re, im = are + y128 + sgn16 * a64, aim
                else:
                    re, im = are - y128, aim - sgn16 * b64
                mg.append(np.sqrt(re ** 2 + im ** 2))                               # [16 channels, 32 streams]
            mg8[w] = mg
        for w in range(4):                                                          # (behind the kernel's barrier 1b)
            m0, m1, m2 = mg8[w]
            for p, val in enumerate((m0, (m0 + m2) + m1, (m0 + m2) - m1, m0 + 2 * m1 + 4 * m2, m2)):
                RX[PS * p + 4 * w:PS * p + 4 * w + 4] = val.reshape(4, 4, 32).transpose(0, 2, 1)   # quad row = channel / 4
    if not k8:
        # 16 kHz: two 16-row tiles per wave on v_mfma_f32_16x16x4_f32 (pack_dft_fold3_wave): tile 0 = 16 odd bins on po | qo (K = 64,
        # blocks ws + 2 j, + 1, j = 0..3), tile 1 = 16 even bins on pe-+ | qe+- (waves 0, 1: m odd) or pe+ | qe- (waves 2, 3), K = 32
        sgn16 = np.where(np.arange(16) % 2 == 0, 1.0, -1.0)[:, None]
        blk = lambda b: W[b].astype(np.float64).reshape(4, 16, 4)                   # [kq][row][i]
        mg16 = {}
        for w in range(4):
            ws = sect[w][S_STFT]
            eR, eI = (40, 56) if w < 2 else (32, 48)
            for rt in range(2):
                mg = []
                for c in range(3):
                    are, aim = np.zeros((16, 32)), np.zeros((16, 32))
                    for j in range(4 if rt == 0 else 2):
                        b0 = ws + (2 * j if rt == 0 else 8 + 2 * j)
                        rR, rI = (4 * j, 16 + 4 * j) if rt == 0 else (eR + 4 * j, eI + 4 * j)
                        are += np.einsum("kri,kmi->rm", blk(b0), RX[CS * c + rR:CS * c + rR + 4])
                        aim += np.einsum("kri,kmi->rm", blk(b0 + 1), RX[CS * c + rI:CS * c + rI + 4])
                    y128, a64, b64 = fcor[c, 0][None, :], fcor[c, 1][None, :], fcor[c, 2][None, :]
                    if rt == 0:
                        re, im = are - y128, aim - sgn16 * b64
                    else:
                        re, im = are + (y128 - a64 if w < 2 else y128 + a64), aim
                    mg.append(np.sqrt(re ** 2 + im ** 2))                           # [16 channels, 32 streams]
                mg16[w, rt] = mg
        for (w, rt), (m0, m1, m2) in mg16.items():                                  # (behind the kernel's barrier 1b)
            for p, val in enumerate((m0, (m0 + m2) + m1, (m0 + m2) - m1, m0 + 2 * m1 + 4 * m2, m2)):
                r0 = PS * p + 8 * w + 4 * rt
                RX[r0:r0 + 4] = val.reshape(4, 4, 32).transpose(0, 2, 1)            # quad row = channel / 4
    n0, n1, n2 = nyq
    RX[ROWN] = np.stack([n0, (n0 + n2) + n1, (n0 + n2) - n1, n0 + 2 * n1 + 4 * n2], axis=1)
    RX[ROWN + 2] = np.stack([n2, np.zeros(32), np.zeros(32), np.zeros(32)], axis=1)
    RX[ROWN + 1] = 0
    RX[ROWN + 3] = 0
    # enc0: five point-wise contractions, then the interpolation (vad_layout.h)
    E0 = {}
    for w in range(4):
        ws = sect[w][S_ENC0]
        bias = np.repeat(_vec(W[ws:ws + 4])[:, None], 32, 1)
        P = [np.zeros((32, 32)) for _ in range(5)]
        ws += 4
        for j in range(NJ0):
            for p in range(5):
                P[p] += _mfma4(W[ws + 5 * j + p], _rows(RX, PS * p + 2 * j, PS * p + 2 * j + 1))
        an, bn = _rows(RX, ROWN, ROWN + 1), _rows(RX, ROWN + 2, ROWN + 3)
        wa, wb = W[ws + 5 * NJ0].astype(np.float64).reshape(2, 32, 4), W[ws + 5 * NJ0 + 1].astype(np.float64).reshape(2, 32, 4)
        for p in range(4):     # one K = 2 MFMA per point: component p of block A against component p of the activation quad
            P[p] += np.einsum("hn,hm->nm", wa[:, :, p], an.reshape(2, 32, 4)[:, :, p])
        P[4] += np.einsum("hn,hm->nm", wb[:, :, 0], bn.reshape(2, 32, 4)[:, :, 0])
        y0, y4 = P[0], P[4]
        bb = P[1] - P[2]
        y2 = (P[1] + P[2]) - y0 - y4
        t2 = (P[3] - y0) - 4 * y2 - 16 * y4
        y3 = t2 / 6 - bb / 3
        E0[w] = [(bb - y3) + bias, y2 + bias, y3 + bias]
    for w in range(4):
        for c in range(3):
            _store_tile(RE, c * 32 + 8 * w, E0[w][c])
    # enc1
    E1 = {}
    for w in range(4):
        nt, tp = w & 1, w >> 1
        ws = sect[w][S_ENC1]
        acc = np.repeat(_vec(W[ws:ws + 4])[:, None], 32, 1)
        ws += 4
        for it in range(32):
            ti, j = it >> 4, it & 15
            r = (tp + ti) * 32 + 2 * j
            acc += _mfma4(W[ws + it], _rows(RE, r, r + 1))
        E1[w] = acc
    for w in range(4):
        _store_tile(RX, (w >> 1) * 16 + 8 * (w & 1), E1[w])
    # enc2
    # split-K over the 4 waves: wave w = tile w&1, K half w>>1; partial tiles un-activated in rows 16 (w>>1) + 8 (w&1)
    E2 = {}
    for w in range(4):
        ws = sect[w][S_ENC2]
        kh = w >> 1
        acc = np.repeat(_vec(W[ws:ws + 4])[:, None], 32, 1) if kh == 0 else np.zeros((32, 32))
        ws += 4
        for it in range(8 * kh, 8 * kh + 8):
            ti, j = it >> 3, it & 7
            r = ti * 16 + 2 * j
            acc += _mfma4(W[ws + it], _rows(RX, r, r + 1))
        E2[w] = acc
    for w in range(4):
        _store_tile(RE, 16 * (w >> 1) + 8 * (w & 1), E2[w], relu=False)
    RE[0:16] = np.maximum(RE[0:16] + RE[16:32], 0.0)          # enc3 reads relu(half 0 + half 1)
    # enc3
    E3 = {}
    for w in range(4):
        ws = sect[w][S_ENC3]
        acc = np.repeat(_vec(W[ws:ws + 4])[:, None], 32, 1)
        ws += 4
        for j in range(8):
            acc += _mfma4(W[ws + j], _rows(RE, 2 * j, 2 * j + 1))
        E3[w] = acc
    for w in range(4):
        _store_tile(RX, 8 * w, E3[w])
    # LSTM + head
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    h_new = np.zeros((32, 128))
    c_new = np.zeros((32, 128))
    z = np.zeros(32)
    for w in range(4):
        ws = sect[w][S_LSTM]
        # the accumulators start from the section's COMPACT bias block (floats [gate][unit]; LSTM_BIAS_BLOCK in vad_layout.h);
        # the lane-expanded copies at the head of the section must say the same
        cb = W[ws + 16 + 64 + 64 + 4].reshape(-1)[:128].reshape(4, 32)
        for q in range(4):
            assert np.array_equal(cb[q], _vec(W[ws + 4 * q:ws + 4 * q + 4]).astype(cb.dtype))
        g = [np.repeat(cb[q].astype(np.float64)[:, None], 32, 1) for q in range(4)]
        ws += 16
        for src in (RX, RH):
            for j in range(16):
                a = _rows(src, 2 * j, 2 * j + 1)
                for q in range(4):
                    g[q] += _mfma4(W[ws + 4 * j + q], a)
            ws += 64
        hw = _vec(W[ws:ws + 4])
        cp = c_prev[:, 32 * w:32 * w + 32].T  # [unit, m]
        cn = sig(g[1]) * cp + sig(g[0]) * np.tanh(g[2])
        hn = sig(g[3]) * np.tanh(cn)
        h_new[:, 32 * w:32 * w + 32] = hn.T
        c_new[:, 32 * w:32 * w + 32] = cn.T
        z += (hw[:, None] * np.maximum(hn, 0)).sum(0)
    hb = W[sect[0][S_HEADB]][0, 0]
    prob = sig(z + hb)
    return prob.astype(np.float32), np.concatenate([h_new, c_new], axis=1).astype(np.float32)


# ======================================================================================
#  Silero V4: model of silero_v4_step (STFT part + tail) over the packed streams
# ======================================================================================
V4 = dict(S_STFT=0, S_NYQ=1, S_DW0=2, S_L0=3, S_S0=4, S_L1=5, S_S1=6, S_L2=7, S_S2=8, S_L3=9, S_S3=10, S_LSTM0=11,
          S_LSTM1=12, S_HEADB=13)


def _table_row(W, blk, row):
    """float4 row `row` of a VALU table that starts at block `blk` (16 bytes per row)."""
    return W[blk:].reshape(-1, 4)[row].astype(np.float64)


def _bias_tile(W, blk):
    return np.repeat(_vec(W[blk:blk + 4])[:, None], 32, 1)


def v4_step(W, sect, x, hc, gate=0.01):
    """x [32,512], hc [32,256] (h0 h1 c0 c1) -> (prob [32], new hc)."""
    x = x.astype(np.float64)
    if gate is not None and gate >= 0:
        x = np.where(np.abs(x) > gate, x, 0.0)
    S = V4
    # ---- STFT part: reflect pad, fold, STFT, magnitudes (registers in the kernel) -> rows 33 t + q of the tail's LDS layout
    xp = np.pad(x, ((0, 0), (96, 96)), mode="reflect")                    # [32, 704]
    mags = np.zeros((264, 32, 4))
    wtab = W[sect[0][S["S_NYQ"]]].reshape(-1)[:256].astype(np.float64)    # w[n], the k = 0 row of the stored basis
    n = np.arange(64)
    sgn = np.where(np.arange(32) % 2 == 0, 1.0, -1.0)[:, None]             # (-1)^r per tile row
    for grp in range(4):
        UV = np.zeros((128, 32, 4))
        fcor = np.zeros((2, 3, 32))
        for cp in range(2):
            t = 2 * grp + cp
            y = xp[:, 64 * t:64 * t + 256] * wtab[None, :]
            y1, y2, y3, y4 = y[:, n], y[:, 128 - n], y[:, 128 + n], y[:, (256 - n) % 256]
            pe, po = y1 + y4 + y2 + y3, y1 + y4 - y2 - y3
            qe, qo = y1 - y4 - y2 + y3, y1 - y4 + y2 - y3
            for k, arr in enumerate((pe, po, qe, qo)):
                arr = arr.copy()
                arr[:, 0] = 0.0                                             # n = 0 is not part of the folded sums
                UV[64 * cp + 16 * k:64 * cp + 16 * k + 16] = arr.reshape(32, 16, 4).transpose(1, 0, 2)
            fcor[cp, 0], fcor[cp, 1], fcor[cp, 2] = y[:, 128], y[:, 64] + y[:, 192], y[:, 64] - y[:, 192]
            pe4 = UV[64 * cp:64 * cp + 16]
            alt = (pe4[:, :, 0] - pe4[:, :, 1] + pe4[:, :, 2] - pe4[:, :, 3]).sum(0)
            mags[33 * t + 32, :, 0] = np.abs(alt + fcor[cp, 0] + fcor[cp, 1])
        for w in range(4):
            ws = sect[w][S["S_STFT"]]
            rR, rI = (0, 32) if w < 2 else (16, 48)
            are = [np.zeros((32, 32)) for _ in range(2)]
            aim = [np.zeros((32, 32)) for _ in range(2)]
            for j in range(8):
                for cp in range(2):
                    are[cp] += _mfma4(W[ws + 2 * j], _rows(UV, 64 * cp + rR + 2 * j, 64 * cp + rR + 2 * j + 1))
                    aim[cp] += _mfma4(W[ws + 2 * j + 1], _rows(UV, 64 * cp + rI + 2 * j, 64 * cp + rI + 2 * j + 1))
            for cp in range(2):
                y128, a64, b64 = fcor[cp, 0][None, :], fcor[cp, 1][None, :], fcor[cp, 2][None, :]
                if w < 2:
                    re, im = are[cp] + y128 + sgn * a64, aim[cp]
                else:
                    re, im = are[cp] - y128, aim[cp] - sgn * b64
                _store_tile(mags, 33 * (2 * grp + cp) + 8 * w, np.sqrt(re ** 2 + im ** 2), relu=False)
    # ---- tail
    RX = np.zeros((280, 32, 4))
    RX[:264] = mags
    lg = lambda mg: np.log(1.0 + mg * 1048576.0)
    o_dw0 = sect[0][S["S_DW0"]]
    colmean = np.zeros((8, 32))
    for t in range(8):
        sp = lg(RX[33 * t:33 * t + 33])                                     # [33, 32, 4]
        colmean[t] = (sp[:32].sum(axis=(0, 2)) + sp[32, :, 0]) / 129.0
    f0, f1 = _table_row(W, o_dw0, 2 * 34 * 6), _table_row(W, o_dw0, 2 * 34 * 6 + 1)
    filt = np.concatenate([f0, f1[:3]])
    mp = np.concatenate([colmean[[3, 2, 1]], colmean, colmean[[6, 5, 4]]])  # [14, 32]
    mm = np.mean([sum(filt[k] * mp[t + k] for k in range(7)) for t in range(8)], axis=0)   # [32]
    # P2 first layer: 16x16x4 tiles, K split over the waves (j = w, w + 4), Nyquist channel as a rank-1 term of wave w
    o_l0 = sect[0][S["S_L0"]]
    ws = o_l0 + 5

    def vec16(blk):                               # vector_block16: lane (n, rq) holds channels 4 rq .. + 3
        return np.concatenate([W[blk][16 * rq].astype(np.float64) for rq in range(4)])

    def mfma16(wblk, a):                          # wblk lane (r, kq), a lane (n, kq): D[r][n]
        return np.einsum("kri,kni->rn", wblk.astype(np.float64).reshape(4, 16, 4), a.reshape(4, 16, 4))

    part = np.zeros((4, 4, 16, 32))               # [wave][column][channel][stream]
    part[0] += vec16(o_l0)[None, :, None]
    wn = [vec16(o_l0 + 1 + k) for k in range(4)]
    for w in range(4):
        for it in range(2):
            j = w + 4 * it
            for sg in range(2):
                st = slice(16 * sg, 16 * sg + 16)
                for c in range(4):
                    frag = {k: np.zeros((64, 4)) for k in ("dm", "xm", "dn", "xn")}
                    for kq in range(4):
                        q = 4 * j + kq
                        dm = np.repeat(_table_row(W, o_dw0, q * 6 + 5)[None], 16, 0)
                        dn = np.repeat(_table_row(W, o_dw0, (34 + q) * 6 + 5)[None], 16, 0)
                        for k in range(5):
                            tc = 2 * c + k - 2
                            if 0 <= tc < 8:
                                mg = RX[33 * tc + q][st]
                                sp = lg(mg) - mm[st, None]
                                dm = dm + _table_row(W, o_dw0, q * 6 + k)[None] * mg
                                dn = dn + _table_row(W, o_dw0, (34 + q) * 6 + k)[None] * sp
                        sl = slice(16 * kq, 16 * kq + 16)
                        mg0 = RX[33 * 2 * c + q][st]
                        frag["dm"][sl], frag["xm"][sl] = np.maximum(dm, 0), mg0
                        frag["dn"][sl], frag["xn"][sl] = np.maximum(dn, 0), lg(mg0) - mm[st, None]
                    for i, k in enumerate(("dm", "xm", "dn", "xn")):
                        part[w, c, :, st] += mfma16(W[ws + 4 * j + i], frag[k])
        # Nyquist channel (table quad 32, component 0) of output column w
        dm = np.full(32, _table_row(W, o_dw0, 32 * 6 + 5)[0])
        dn = np.full(32, _table_row(W, o_dw0, (34 + 32) * 6 + 5)[0])
        for k in range(5):
            tc = 2 * w + k - 2
            if 0 <= tc < 8:
                mg = RX[33 * tc + 32][:, 0]
                dm = dm + _table_row(W, o_dw0, 32 * 6 + k)[0] * mg
                dn = dn + _table_row(W, o_dw0, (34 + 32) * 6 + k)[0] * (lg(mg) - mm)
        xm = RX[33 * 2 * w + 32][:, 0]
        xn = lg(xm) - mm
        part[w, w] += (np.outer(wn[0], np.maximum(dm, 0)) + np.outer(wn[1], xm) + np.outer(wn[2], np.maximum(dn, 0)) +
                       np.outer(wn[3], xn))
    first = np.maximum(part.sum(axis=0), 0)       # [column][16 channels][32 streams]
    for c in range(4):
        for q in range(4):
            RX[264 + 4 * c + q] = first[c, 4 * q:4 * q + 4].T
    hprev = hc[:, :128].astype(np.float64)
    RX[120:152] = hprev.reshape(32, 32, 4).transpose(1, 0, 2)

    def store16(row0, acc):                       # rows 0..15 of the tile only
        v = np.maximum(acc, 0)
        for g in range(2):
            for h in range(2):
                RX[row0 + 2 * g + h] = v[8 * g + 4 * h:8 * g + 4 * h + 4].T

    def dwq(tab_blk, q, taps):
        """taps: list of (k, quad [32,4])"""
        d = np.repeat(_table_row(W, tab_blk, q * 6 + 5)[None], 32, 0)
        for k, quad in taps:
            d = d + _table_row(W, tab_blk, q * 6 + k)[None] * quad
        return np.maximum(d, 0)

    # P3 s0
    o = sect[0][S["S_S0"]]
    res = {}
    for w in range(4):
        acc = _bias_tile(W, o)
        acc += _mfma4(W[o + 4], _rows(RX, 264 + 4 * w, 264 + 4 * w + 1))
        acc += _mfma4(W[o + 5], _rows(RX, 264 + 4 * w + 2, 264 + 4 * w + 3))
        res[w] = acc
    for w in range(4):
        store16(0 + 4 * w, res[w])
    # P4 block 1
    o = sect[0][S["S_L1"]]
    res = {}
    for w in range(4):
        acc = _bias_tile(W, o + 1)
        d, y = [], []
        for j in range(2):
            fd = np.zeros((64, 4))
            for h in range(2):
                q = 2 * j + h
                taps = [(k, RX[0 + 4 * (w + k - 2) + q]) for k in range(5) if 0 <= w + k - 2 < 4]
                fd[32 * h:32 * h + 32] = dwq(o, q, taps)
            d.append(fd)
            y.append(_rows(RX, 0 + 4 * w + 2 * j, 0 + 4 * w + 2 * j + 1))
        for i, fr in enumerate(d + y):
            acc += _mfma4(W[o + 5 + i], fr)
        res[w] = acc
    for w in range(4):
        _store_tile(RX, 16 + 8 * w, res[w])
    # P5 s1 (columns 0 and 2)
    o = sect[0][S["S_S1"]]
    res = {}
    for w in range(2):
        acc = _bias_tile(W, o)
        r = 16 + 8 * (2 * w)
        for j in range(4):
            acc += _mfma4(W[o + 4 + j], _rows(RX, r + 2 * j, r + 2 * j + 1))
        res[w] = acc
    for w in range(2):
        _store_tile(RX, 48 + 8 * w, res[w])
    # P6 block 2 (identity residual)
    o = sect[0][S["S_L2"]]
    res = {}
    for w in range(2):
        acc = _bias_tile(W, o + 1)
        for j in range(4):
            fd = np.zeros((64, 4))
            for h in range(2):
                q = 2 * j + h
                taps = [(k, RX[48 + 8 * (w + k - 2) + q]) for k in range(5) if 0 <= w + k - 2 < 2]
                fd[32 * h:32 * h + 32] = dwq(o, q, taps)
            acc += _mfma4(W[o + 5 + j], fd)
        resid = np.zeros((32, 32))
        for g in range(4):
            for h in range(2):
                resid[8 * g + 4 * h:8 * g + 4 * h + 4] = RX[48 + 8 * w + 2 * g + h].T
        res[w] = acc + resid
    for w in range(2):
        _store_tile(RX, 64 + 8 * w, res[w])
    # P7 s2 (column 0)
    o = sect[0][S["S_S2"]]
    acc = _bias_tile(W, o)
    for j in range(4):
        acc += _mfma4(W[o + 4 + j], _rows(RX, 64 + 2 * j, 64 + 2 * j + 1))
    _store_tile(RX, 80, acc)
    # P8 block 3
    o = sect[0][S["S_L3"]]
    res = {}
    for w in range(2):
        ob = o + 1 + 12 * w
        acc = _bias_tile(W, ob)
        for j in range(4):
            fd = np.zeros((64, 4))
            for h in range(2):
                q = 2 * j + h
                fd[32 * h:32 * h + 32] = dwq(o, q, [(2, RX[80 + q])])
            acc += _mfma4(W[ob + 4 + j], fd)
            acc += _mfma4(W[ob + 8 + j], _rows(RX, 80 + 2 * j, 80 + 2 * j + 1))
        res[w] = acc
    for w in range(2):
        _store_tile(RX, 88 + 8 * w, res[w])
    # P9 s3
    o = sect[0][S["S_S3"]]
    res = {}
    for w in range(2):
        ob = o + 12 * w
        acc = _bias_tile(W, ob)
        for j in range(8):
            acc += _mfma4(W[ob + 4 + j], _rows(RX, 88 + 2 * j, 88 + 2 * j + 1))
        res[w] = acc
    for w in range(2):
        _store_tile(RX, 104 + 8 * w, res[w])
    # LSTMs
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    new = hc.astype(np.float64).copy()
    z = np.zeros(32)
    oh = sect[0][S["S_HEADB"]]
    for layer in range(2):
        xin = 104 if layer == 0 else 152
        hin = 120 if layer == 0 else 136
        hn_all = np.zeros((64, 32))
        for w in range(4):                                   # wave w: units 16w..16w+15, tiles A = i|f, B = g|o
            ob = sect[w][S["S_LSTM0" if layer == 0 else "S_LSTM1"]]
            gA, gB = _bias_tile(W, ob), _bias_tile(W, ob + 4)
            for it in range(16):
                src = (xin if it < 8 else hin) + 2 * (it & 7)
                gA += _mfma4(W[ob + 8 + 2 * it], _rows(RX, src, src + 1))
                gB += _mfma4(W[ob + 8 + 2 * it + 1], _rows(RX, src, src + 1))
            gi, gf, gg, go = gA[:16], gA[16:], gB[:16], gB[16:]
            sl = slice(16 * w, 16 * w + 16)
            cp = hc[:, 128 + 64 * layer:192 + 64 * layer].astype(np.float64).T[sl]
            cn = sig(gf) * cp + sig(gi) * np.tanh(gg)
            hn = sig(go) * np.tanh(cn)
            new[:, 128 + 64 * layer + 16 * w:128 + 64 * layer + 16 * w + 16] = cn.T
            new[:, 64 * layer + 16 * w:64 * layer + 16 * w + 16] = hn.T
            hn_all[sl] = hn
            if layer == 1:
                hw = _vec(W[oh + 1 + 4 * (w >> 1):oh + 5 + 4 * (w >> 1)])[16 * (w & 1):16 * (w & 1) + 16]
                z += (hw[:, None] * np.maximum(hn, 0)).sum(0)
        if layer == 0:
            RX[152:168] = hn_all.T.reshape(32, 16, 4).transpose(1, 0, 2)
    prob = sig(z + W[oh][0, 0])
    return prob.astype(np.float32), new.astype(np.float32)


def resample_512(x):
    """NumPy model of ``vadk_resample_512`` (csrc/resample.hip): the four folded inputs, the packed operator stream read
    block by block as the waves read it, the rank-1 accumulator init, the VALU row and the epilogue's recombination.
    x [n <= 32, n_in] -> y [n, 512] (float64 arithmetic)."""
    lib = _ffi.lib()
    n, n_in = x.shape
    nf, tb, r128 = C.c_size_t(), C.c_uint32(), C.c_uint32()
    assert lib.vad_debug_pack_resample(n_in, None, 0, C.byref(nf), C.byref(tb), C.byref(r128)) == 0
    flat = np.empty(nf.value, np.float32)
    assert lib.vad_debug_pack_resample(n_in, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size, C.byref(nf), C.byref(tb),
                                       C.byref(r128)) == 0
    W = flat.reshape(-1, 64, 4)
    tb, r128 = tb.value, r128.value
    Q, H = n_in // 4, n_in // 2
    xs = np.zeros((MT, n_in))
    xs[:n] = x
    # store_chunk: folded quads, j = 4q + e
    j = np.arange(Q)
    a, c = xs[:, j], xs[:, j + H]
    b, d = xs[:, (H - j) % n_in], xs[:, (n_in - j) % n_in]
    pe, me, qe, qo = a + c, a - c, b + d, b - d
    ops = [pe + qe, pe - qe, me + qo, me - qo]                  # ue, ve, uo, vo  [32, Q]
    ops[0][:, 0], ops[1][:, 0], ops[2][:, 0], ops[3][:, 0] = pe[:, 0], 0.0, 0.0, me[:, 0]
    quads = [o.reshape(MT, Q // 4, 4).transpose(1, 0, 2) for o in ops]      # [quad row][m][4]
    mids = [xs[:, Q] + xs[:, 3 * Q], xs[:, Q] - xs[:, 3 * Q]]
    y = np.zeros((MT, 512))
    for rt in range(4):
        base = rt * tb
        acc = [np.outer(_vec(W[base:base + 4]), mids[0]), np.zeros((32, MT)),
               np.outer(_vec(W[base + 4:base + 8]), mids[1]), np.zeros((32, MT))]
        for kj in range(Q // 8):
            for p in range(4):
                acc[p] += _mfma4(W[base + 8 + 4 * kj + p], _rows(quads[p], 2 * kj, 2 * kj + 1))
        se, ae, so, ao = acc
        for r in range(32):
            o = 32 * rt + r
            y[:, o] = se[r] + ae[r] + so[r] + ao[r]
            y[:, o + 256] = se[r] + ae[r] - so[r] - ao[r]
            if o:
                y[:, 256 - o] = se[r] - ae[r] + so[r] - ao[r]
                y[:, 512 - o] = se[r] - ae[r] - so[r] + ao[r]
    row = W[r128:].reshape(-1)
    e = ops[0] @ row[:Q].astype(np.float64) + row[2 * Q] * mids[0]
    od = ops[2] @ row[Q:2 * Q].astype(np.float64) + row[2 * Q + 1] * mids[1]
    y[:, 128], y[:, 384] = e + od, e - od
    return y[:n]


# ======================================================================================
#  Silero V5 on 16-stream tiles: model of silero_v5_t16.hip over pack_silero_v5_t16's streams
# ======================================================================================
def _mfma16(wblk, a):
    """wblk [64,4] (lane = kq*16 + row), a [64,4] (lane = kq*16 + m) -> D[row, m] contribution (K = 16 channels)."""
    w = wblk.astype(np.float64).reshape(4, 16, 4)
    x = a.astype(np.float64).reshape(4, 16, 4)
    return np.einsum("kri,kmi->rm", w, x)


def _rows16(region, base):
    """activation fragment of one k-iteration: lane (m, kq) reads quad row base + kq of stream m"""
    return np.concatenate([region[base + k] for k in range(4)], axis=0)      # [64, 4]


def _vec16(block):
    """one D-layout vector block -> per-channel vector [16] (channel = 4 rq + i)"""
    return block.astype(np.float64).reshape(4, 16, 4)[:, 0, :].reshape(16)


def _store16(region, row0, acc, relu=True):
    v = np.maximum(acc, 0) if relu else acc
    for rq in range(4):
        region[row0 + rq] = v[4 * rq:4 * rq + 4].T                             # [16 m, 4]


def v5_step_t16(W, sect, x, hc, gate=0.01, k8=False):
    """x [16,512] f32, hc [16,256] -> (prob [16], new hc [16,256]); float64 contractions; mirrors silero_v5_t16.hip.
    k8: the 8 kHz sub-model instantiation (x [16,256]; window 128, hop 64, four 16-row STFT tiles, planes of 16 rows)."""
    ROW_NYQ, ROW_E = (80 if k8 else 160), 168
    N = 128 if k8 else 256
    H, Q4 = N // 2, N // 4
    QL = Q4 // 4
    CS, PS, NJ0 = 4 * QL, (16 if k8 else 32), (4 if k8 else 8)
    x = x.astype(np.float64)
    if gate is not None and gate >= 0:
        x = np.where(np.abs(x) > gate, x, 0.0)
    RX = np.zeros((264, 16, 4))
    RE = RX[ROW_E:]
    RH = hc[:, :128].astype(np.float64).reshape(16, 32, 4).transpose(1, 0, 2).copy()
    c_prev = hc[:, 128:].astype(np.float64)
    wtab = W[sect[0][S_NYQ]].reshape(-1)[:N].astype(np.float64)
    fcor = np.zeros((3, 3, 16))
    for c in range(3):
        y = x[:, H * c:H * c + N] * wtab[None, :]
        n = np.arange(Q4)
        y1, y2, y3, y4 = y[:, n], y[:, H - n], y[:, H + n], y[:, (N - n) % N]
        pe, po, qe, qo = y1 + y4 + y2 + y3, y1 + y4 - y2 - y3, y1 - y4 - y2 + y3, y1 - y4 + y2 - y3
        for arr in (pe, po, qe, qo):
            arr[:, 0] = 0.0
        if k8:
            for k, arr in enumerate((pe, po, qe, qo)):
                RX[CS * c + QL * k:CS * c + QL * k + QL] = arr.reshape(16, QL, 4).transpose(1, 0, 2)
        else:
            mm = np.arange(32)                              # the even bins' operands fold once more about n = 32 (silero_v5.hip)
            pep, pen = pe[:, mm] + pe[:, (64 - mm) % 64], pe[:, mm] - pe[:, (64 - mm) % 64]
            qen, qep = qe[:, mm] - qe[:, (64 - mm) % 64], qe[:, mm] + qe[:, (64 - mm) % 64]
            pep[:, 0], pen[:, 0], qen[:, 0], qep[:, 0] = pe[:, 32], 0.0, 0.0, qe[:, 32]
            RX[64 * c:64 * c + 16] = po.reshape(16, 16, 4).transpose(1, 0, 2)
            RX[64 * c + 16:64 * c + 32] = qo.reshape(16, 16, 4).transpose(1, 0, 2)
            for k, arr in enumerate((pep, pen, qen, qep)):
                RX[64 * c + 32 + 8 * k:64 * c + 40 + 8 * k] = arr.reshape(16, 8, 4).transpose(1, 0, 2)
        fcor[c] = y[:, H], y[:, Q4] + y[:, H + Q4], y[:, Q4] - y[:, H + Q4]
    nyq = np.zeros((3, 16))
    for c in range(3):
        pe = RX[CS * c:CS * c + QL] if k8 else RX[64 * c + 32:64 * c + 40]      # 16 kHz: the pe+ rows, slot 0 = pe[32]
        nyq[c] = np.abs((pe[:, :, 0] - pe[:, :, 1] + pe[:, :, 2] - pe[:, :, 3]).sum(0) + fcor[c, 0] + fcor[c, 1])
    sgn = np.where(np.arange(16) % 2 == 0, 1.0, -1.0)[:, None]
    mags = {}
    for w in range(4):
        ws = sect[w][S_STFT]
        for c in range(3):
            y128, a64, b64 = fcor[c, 0][None, :], fcor[c, 1][None, :], fcor[c, 2][None, :]
            if k8:          # one 16-row tile per wave: waves 0 / 1 even bins (pe | qe), 2 / 3 odd bins (po | qo); K = 32
                even = w < 2
                rR, rI = (0, 2 * QL) if even else (QL, 3 * QL)
                are, aim = np.zeros((16, 16)), np.zeros((16, 16))
                for j in range(2):
                    are += _mfma16(W[ws + 2 * j], _rows16(RX, CS * c + rR + 4 * j))
                    aim += _mfma16(W[ws + 2 * j + 1], _rows16(RX, CS * c + rI + 4 * j))
                re, im = (are + y128 + sgn * a64, aim) if even else (are - y128, aim - sgn * b64)
                mags[w, c, 0] = np.sqrt(re ** 2 + im ** 2)
                continue
            eR, eI = (40, 56) if w < 2 else (32, 48)
            are, aim = np.zeros((16, 16)), np.zeros((16, 16))
            for j in range(4):                              # row tile 0: 16 odd bins on po | qo, K = 64
                are += _mfma16(W[ws + 2 * j], _rows16(RX, 64 * c + 4 * j))
                aim += _mfma16(W[ws + 2 * j + 1], _rows16(RX, 64 * c + 16 + 4 * j))
            mags[w, c, 0] = np.sqrt((are - y128) ** 2 + (aim - sgn * b64) ** 2)
            are, aim = np.zeros((16, 16)), np.zeros((16, 16))
            for j in range(2):                              # row tile 1: 16 even bins, K = 32
                are += _mfma16(W[ws + 8 + 2 * j], _rows16(RX, 64 * c + eR + 4 * j))
                aim += _mfma16(W[ws + 8 + 2 * j + 1], _rows16(RX, 64 * c + eI + 4 * j))
            mags[w, c, 1] = np.sqrt((are + (y128 - a64 if w < 2 else y128 + a64)) ** 2 + aim ** 2)
    for w in range(4):
        for rt in range(1 if k8 else 2):
            m0, m1, m2 = (mags[w, c, rt] for c in range(3))
            for p, v in enumerate((m0, (m0 + m2) + m1, (m0 + m2) - m1, m0 + 2 * m1 + 4 * m2, m2)):
                _store16(RX, PS * p + (4 * w if k8 else 8 * w + 4 * rt), v, relu=False)
    n0, n1, n2 = nyq
    RX[ROW_NYQ:ROW_NYQ + 8] = 0
    RX[ROW_NYQ] = np.stack([n0, (n0 + n2) + n1, (n0 + n2) - n1, n0 + 2 * n1 + 4 * n2], axis=1)
    RX[ROW_NYQ + 4] = np.stack([n2, np.zeros(16), np.zeros(16), np.zeros(16)], axis=1)
    E0 = {}
    for w in range(4):
        ws = sect[w][S_ENC0]
        P = [[np.zeros((16, 16)) for _ in range(2)] for _ in range(5)]
        for j in range(NJ0):
            for p in range(5):
                a = _rows16(RX, PS * p + 4 * j)
                for rt in range(2):
                    P[p][rt] += _mfma16(W[ws + 2 + 10 * j + 2 * p + rt], a)
        an, bn = _rows16(RX, ROW_NYQ), _rows16(RX, ROW_NYQ + 4)
        for rt in range(2):
            wa = W[ws + 2 + 10 * NJ0 + rt].astype(np.float64).reshape(4, 16, 4)
            wb = W[ws + 2 + 10 * NJ0 + 2 + rt].astype(np.float64).reshape(4, 16, 4)
            for p in range(4):
                P[p][rt] += np.einsum("kr,km->rm", wa[:, :, p], an.reshape(4, 16, 4)[:, :, p])
            P[4][rt] += np.einsum("kr,km->rm", wb[:, :, 0], bn.reshape(4, 16, 4)[:, :, 0])
            bias = np.repeat(_vec16(W[ws + rt])[:, None], 16, 1)
            y0, y4 = P[0][rt], P[4][rt]
            bb = P[1][rt] - P[2][rt]
            y2 = (P[1][rt] + P[2][rt]) - y0 - y4
            y3 = ((P[3][rt] - y0) - 4 * y2 - 16 * y4) / 6 - bb / 3
            E0[w, rt] = [(bb - y3) + bias, y2 + bias, y3 + bias]
    for (w, rt), cols in E0.items():
        for c in range(3):
            _store16(RE, 32 * c + 8 * w + 4 * rt, cols[c])
    E1 = {}
    for w in range(4):
        nt, tp = w & 1, w >> 1
        ws = sect[w][S_ENC1]
        for rt in range(2):
            acc = np.repeat(_vec16(W[ws + rt])[:, None], 16, 1)
            for it in range(16):
                acc += _mfma16(W[ws + 2 + 2 * it + rt], _rows16(RE, (tp + (it >> 3)) * 32 + 4 * (it & 7)))
            E1[w, rt] = acc
    for (w, rt), acc in E1.items():
        _store16(RX, 16 * (w >> 1) + 8 * (w & 1) + 4 * rt, acc)
    E2 = {}
    for w in range(4):
        kh = w >> 1
        ws = sect[w][S_ENC2]
        for rt in range(2):
            acc = np.repeat(_vec16(W[ws + rt])[:, None], 16, 1) if kh == 0 else np.zeros((16, 16))
            for j in range(4):
                acc += _mfma16(W[ws + 2 + 8 * kh + 2 * j + rt], _rows16(RX, 16 * kh + 4 * j))
            E2[w, rt] = acc
    for (w, rt), acc in E2.items():
        _store16(RE, 16 * (w >> 1) + 8 * (w & 1) + 4 * rt, acc, relu=False)
    RE[0:16] = np.maximum(RE[0:16] + RE[16:32], 0.0)
    E3 = {}
    for w in range(4):
        ws = sect[w][S_ENC3]
        for rt in range(2):
            acc = np.repeat(_vec16(W[ws + rt])[:, None], 16, 1)
            for j in range(4):
                acc += _mfma16(W[ws + 2 + 2 * j + rt], _rows16(RE, 4 * j))
            E3[w, rt] = acc
    for (w, rt), acc in E3.items():
        _store16(RX, 8 * w + 4 * rt, acc)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    h_new, c_new, z = np.zeros((16, 128)), np.zeros((16, 128)), np.zeros(16)
    for w in range(4):
        ws = sect[w][S_LSTM]
        # the accumulators start from the section's COMPACT bias block (floats [gate][unit]; T_LSTM_BIAS_BLOCK in silero_v5_t16.hip);
        # the lane-expanded copies at the head of the section must say the same
        cb = W[ws + 8 + 64 + 64 + 2].reshape(-1)[:128].reshape(4, 32)
        for rt in range(2):
            for q in range(4):
                assert np.array_equal(cb[q, 16 * rt:16 * rt + 16], _vec16(W[ws + 2 * q + rt]).astype(cb.dtype))
            g = [np.repeat(cb[q, 16 * rt:16 * rt + 16].astype(np.float64)[:, None], 16, 1) for q in range(4)]
            for half, src in enumerate((RX, RH)):
                for j in range(8):
                    a = _rows16(src, 4 * j)
                    for q in range(4):
                        g[q] += _mfma16(W[ws + 8 + 64 * half + 8 * j + 2 * q + rt], a)
            hw = _vec16(W[ws + 8 + 128 + rt])
            u0 = 32 * w + 16 * rt
            cn = sig(g[1]) * c_prev[:, u0:u0 + 16].T + sig(g[0]) * np.tanh(g[2])
            hn = sig(g[3]) * np.tanh(cn)
            h_new[:, u0:u0 + 16], c_new[:, u0:u0 + 16] = hn.T, cn.T
            z += (hw[:, None] * np.maximum(hn, 0)).sum(0)
    prob = sig(z + W[sect[0][S_HEADB]][0, 0])
    return prob.astype(np.float32), np.concatenate([h_new, c_new], axis=1).astype(np.float32)


def packed_resample_t16(n_in: int):
    lib = _ffi.lib()
    n, wb, r128 = C.c_size_t(), C.c_uint32(), C.c_uint32()
    assert lib.vad_debug_pack_resample_t16(n_in, None, 0, C.byref(n), C.byref(wb), C.byref(r128)) == 0
    out = np.empty(n.value, np.float32)
    assert lib.vad_debug_pack_resample_t16(n_in, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(n), C.byref(wb),
                                           C.byref(r128)) == 0
    return out.reshape(-1, 64, 4), int(wb.value), int(r128.value)


def _resample_t16_up2(W, wave_blocks, x):
    """8 kHz chunks (part shape 3 of the RS prologue): the even outputs are the input samples, copied by the loader threads; per wave
    ONE 16-row tile - the odd rows 32 w + 2 r + 1 - is contracted: two vector blocks (RE, RO at the unpaired sample), then four
    weight blocks (se, ae, so, ao) per k-iteration of 16 folded samples."""
    n_in = 256
    H, Q = 128, 64
    xe, xo = x[:, :H] + x[:, H:], x[:, :H] - x[:, H:]
    j = np.arange(Q)
    rev = (H - j) % H
    ue, ve, uo, vo = xe[:, j] + xe[:, rev], xe[:, j] - xe[:, rev], xo[:, j] + xo[:, rev], xo[:, j] - xo[:, rev]
    ue[:, 0], ve[:, 0], uo[:, 0], vo[:, 0] = xe[:, 0], 0.0, 0.0, xo[:, 0]
    parts = [a.reshape(16, Q // 4, 4).transpose(1, 0, 2) for a in (ue, ve, uo, vo)]
    y = np.zeros((16, 512))
    y[:, 0::2] = x                                                       # the loader's copies
    for w in range(4):
        wb = w * wave_blocks
        acc = [np.zeros((16, 16)) for _ in range(4)]
        acc[0] += _vec16(W[wb])[:, None] * xe[:, Q][None, :]
        acc[2] += _vec16(W[wb + 1])[:, None] * xo[:, Q][None, :]
        for g in range(Q // 16):
            for p in range(4):
                acc[p] += _mfma16(W[wb + 2 + 4 * g + p], _rows16(parts[p], 4 * g))
        se, ae, so, ao = acc
        for r in range(16):
            o = 32 * w + 2 * r + 1
            y[:, o] = (se + ae + so + ao)[r]
            y[:, o + 256] = (se + ae - so - ao)[r]
            y[:, 256 - o] = (se - ae + so - ao)[r]
            y[:, 512 - o] = (se - ae - so + ao)[r]
    return y


def resample_t16(W, wave_blocks, row128_block, x):
    """x [16, n_in] -> y [16, 512]: the RS prologue of silero_v5_t16.hip over pack_resample_operator_t16's stream (float64).
    24 / 48 kHz chunks ("P3": the stream's contraction length is n_in / 6, not n_in / 4): every third input sample is copied
    (scaled) to its output instant and enters one alternating sum; the MFMAs contract the folded samples j = 1, 2, 4, 5, 7, ..."""
    x = x.astype(np.float64)
    if x.shape[1] == 256 and wave_blocks == 2 + 4 * 4:
        return _resample_t16_up2(W, wave_blocks, x)
    n_in = x.shape[1]
    H, Q = n_in // 2, n_in // 4
    Kc = (wave_blocks - 4) * 2
    p3 = Kc != Q
    assert Kc == (2 * Q // 3 if p3 else Q)
    xe, xo = x[:, :H] + x[:, H:], x[:, :H] - x[:, H:]
    j = np.arange(Q)
    rev = (H - j) % H
    ue, ve, uo, vo = xe[:, j] + xe[:, rev], xe[:, j] - xe[:, rev], xo[:, j] + xo[:, rev], xo[:, j] - xo[:, rev]
    ue[:, 0], ve[:, 0], uo[:, 0], vo[:, 0] = xe[:, 0], 0.0, 0.0, xo[:, 0]
    if p3:
        i = np.arange(Kc)
        jm = 3 * (i >> 1) + 1 + (i & 1)
        ue, ve, uo, vo = ue[:, jm], ve[:, jm], uo[:, jm], vo[:, jm]
    parts = [a.reshape(16, Kc // 4, 4).transpose(1, 0, 2) for a in (ue, ve, uo, vo)]      # quad rows [Kc/4, 16 streams, 4]
    y = np.zeros((16, 512))
    for w in range(4):
        wb = w * wave_blocks
        for rt in range(2):
            acc = [np.zeros((16, 16)) for _ in range(4)]
            if not p3:
                acc[0] += _vec16(W[wb + rt])[:, None] * xe[:, Q][None, :]
                acc[2] += _vec16(W[wb + 2 + rt])[:, None] * xo[:, Q][None, :]
            for g in range(Kc // 16):
                for p in range(4):
                    acc[p] += _mfma16(W[wb + 4 + 8 * g + 2 * p + rt], _rows16(parts[p], 4 * g))
            se, ae, so, ao = acc
            for r in range(16):
                o = 32 * w + 16 * rt + r
                y[:, o] = (se + ae + so + ao)[r]
                y[:, o + 256] = (se + ae - so - ao)[r]
                if o:
                    y[:, 256 - o] = (se - ae + so - ao)[r]
                    y[:, 512 - o] = (se - ae - so + ao)[r]
    row = W[row128_block:].reshape(-1).astype(np.float64)
    e = ue @ row[:Kc]
    od = uo @ row[Kc:2 * Kc]
    if not p3:
        e = e + row[2 * Kc] * xe[:, Q]
        od = od + row[2 * Kc + 1] * xo[:, Q]
    y[:, 128], y[:, 384] = e + od, e - od
    if p3:
        # the loader's side of the chunk: the samples 3 i' go straight to F (scaled), and into the alternating sum
        m = 1536 // n_in
        x0 = x[:, 0::3]
        ip = np.arange(n_in // 3)
        y[:, m * ip] += (512.0 / n_in) * x0
        A = (x0 * np.where((m * ip) % 2 == 0, 1.0, -1.0)[None, :]).sum(1) if m == 1 else x0.sum(1)
        y += (A / n_in)[:, None] * np.where(np.arange(512) % 2 == 0, 1.0, -1.0)[None, :]
    return y


# ======================================================================================
#  Silero V4 on 16-stream tiles: model of silero_v4_t16.hip over pack_silero_v4_t16's streams
# ======================================================================================
def v4_step_t16(W, sect, x, hc, gate=0.01, k8=False):
    """x [16,512] f32, hc [16,256] (h0 h1 c0 c1) -> (prob [16], new hc); float64 contractions; mirrors silero_v4_t16.hip:
    same rows, block offsets and wave roles.  k8: the graph's 8 kHz sub-model (two columns through block 3, two LSTM steps)."""
    S = V4
    x = x.astype(np.float64)
    if gate is not None and gate >= 0:
        x = np.where(np.abs(x) > gate, x, 0.0)
    xp = np.pad(x, ((0, 0), (96, 96)), mode="reflect")                    # [16, 704]
    wtab = W[sect[0][S["S_NYQ"]]].reshape(-1)[:256].astype(np.float64)
    RX = np.zeros((280, 16, 4))
    n = np.arange(64)
    sgn = np.where(np.arange(16) % 2 == 0, 1.0, -1.0)[:, None]
    for grp in range(4):
        UV = np.zeros((128, 16, 4))
        fcor = np.zeros((2, 3, 16))
        for cp in range(2):
            t = 2 * grp + cp
            y = xp[:, 64 * t:64 * t + 256] * wtab[None, :]
            y1, y2, y3, y4 = y[:, n], y[:, 128 - n], y[:, 128 + n], y[:, (256 - n) % 256]
            pe, po, qe, qo = y1 + y4 + y2 + y3, y1 + y4 - y2 - y3, y1 - y4 - y2 + y3, y1 - y4 + y2 - y3
            for arr in (pe, po, qe, qo):
                arr[:, 0] = 0.0
            # the even bins' operands fold once more about n = 32; the unpaired n = 32 rides in slot 0 (pe[32]: k / 2 even, qe[32]: odd)
            m = np.arange(32)
            pee, peo = pe[:, m] + pe[:, (64 - m) % 64], pe[:, m] - pe[:, (64 - m) % 64]
            qee, qeo = qe[:, m] - qe[:, (64 - m) % 64], qe[:, m] + qe[:, (64 - m) % 64]
            pee[:, 0], peo[:, 0], qee[:, 0], qeo[:, 0] = pe[:, 32], 0.0, 0.0, qe[:, 32]
            base = 64 * cp                                                  # rows: po 0..15 | qo 16..31 | pee | peo | qee | qeo (8 each)
            UV[base:base + 16] = po.reshape(16, 16, 4).transpose(1, 0, 2)
            UV[base + 16:base + 32] = qo.reshape(16, 16, 4).transpose(1, 0, 2)
            for k, arr in enumerate((pee, peo, qee, qeo)):
                UV[base + 32 + 8 * k:base + 40 + 8 * k] = arr.reshape(16, 8, 4).transpose(1, 0, 2)
            fcor[cp] = y[:, 128], y[:, 64] + y[:, 192], y[:, 64] - y[:, 192]
            e, o = y[:, 0::2].sum(1), y[:, 1::2].sum(1)                      # the two real bins straight from the samples (float64)
            RX[33 * t + 32, :, 0] = np.abs(e - o)
            dc = e + o
            for w in range(4):
                ws = sect[w][S["S_STFT"]]
                y128, a64, b64 = fcor[cp, 0][None, :], fcor[cp, 1][None, :], fcor[cp, 2][None, :]
                # row tile 0: the wave's 16 odd bins, K = 64
                are, aim = np.zeros((16, 16)), np.zeros((16, 16))
                for j in range(4):
                    are += _mfma16(W[ws + 2 * j], _rows16(UV, base + 4 * j))
                    aim += _mfma16(W[ws + 2 * j + 1], _rows16(UV, base + 16 + 4 * j))
                re, im = are - y128, aim - sgn * b64
                _store16(RX, 33 * t + 8 * w, np.sqrt(re ** 2 + im ** 2), relu=False)
                # row tile 1: 16 even bins, K = 32; waves 0, 1: k / 2 odd (peo | qeo), waves 2, 3: k / 2 even (pee | qee)
                eR, eI = (40, 56) if w < 2 else (32, 48)
                are, aim = np.zeros((16, 16)), np.zeros((16, 16))
                for j in range(2):
                    are += _mfma16(W[ws + 8 + 2 * j], _rows16(UV, base + eR + 4 * j))
                    aim += _mfma16(W[ws + 8 + 2 * j + 1], _rows16(UV, base + eI + 4 * j))
                re, im = are + (y128 - a64 if w < 2 else y128 + a64), aim
                if w == 2:
                    re[0] = dc
                _store16(RX, 33 * t + 8 * w + 4, np.sqrt(re ** 2 + im ** 2), relu=False)
    lg = lambda mg: np.log(1.0 + mg * 1048576.0)
    o_dw0 = sect[0][S["S_DW0"]]
    colmean = np.zeros((8, 16))
    for t in range(8):
        sp = lg(RX[33 * t:33 * t + 33])
        colmean[t] = (sp[:32].sum(axis=(0, 2)) + sp[32, :, 0]) / 129.0
    f0, f1 = _table_row(W, o_dw0, 2 * 34 * 6), _table_row(W, o_dw0, 2 * 34 * 6 + 1)
    filt = np.concatenate([f0, f1[:3]])
    mp = np.concatenate([colmean[[3, 2, 1]], colmean, colmean[[6, 5, 4]]])
    mm = np.mean([sum(filt[k] * mp[t + k] for k in range(7)) for t in range(8)], axis=0)   # [16]
    # P2 first layer (as in the 32-stream kernel, one stream half)
    o_l0 = sect[0][S["S_L0"]]
    ws = o_l0 + 5
    vec = lambda blk: np.repeat(_vec16(W[blk])[:, None], 16, 1)                # bias tile [16 rows, 16 streams]
    part = np.zeros((4, 4, 16, 16))               # [wave][column][channel][stream]
    part[0] += vec(o_l0)[None]
    wn = [_vec16(W[o_l0 + 1 + k]) for k in range(4)]
    for w in range(4):
        for it in range(2):
            j = 2 * w + it                         # the bins the wave produced itself (registers in the kernel)
            for c in range(4):
                frag = {k: np.zeros((64, 4)) for k in ("dm", "xm", "dn", "xn")}
                for kq in range(4):
                    q = 4 * j + kq
                    dm = np.repeat(_table_row(W, o_dw0, q * 6 + 5)[None], 16, 0)
                    dn = np.repeat(_table_row(W, o_dw0, (34 + q) * 6 + 5)[None], 16, 0)
                    for k in range(5):
                        tc = 2 * c + k - 2
                        if 0 <= tc < 8:
                            mg = RX[33 * tc + q]
                            dm = dm + _table_row(W, o_dw0, q * 6 + k)[None] * mg
                            dn = dn + _table_row(W, o_dw0, (34 + q) * 6 + k)[None] * (lg(mg) - mm[:, None])
                    sl = slice(16 * kq, 16 * kq + 16)
                    mg0 = RX[33 * 2 * c + q]
                    frag["dm"][sl], frag["xm"][sl] = np.maximum(dm, 0), mg0
                    frag["dn"][sl], frag["xn"][sl] = np.maximum(dn, 0), lg(mg0) - mm[:, None]
                for i, k in enumerate(("dm", "xm", "dn", "xn")):
                    part[w, c] += _mfma16(W[ws + 4 * j + i], frag[k])
        dm = np.full(16, _table_row(W, o_dw0, 32 * 6 + 5)[0])
        dn = np.full(16, _table_row(W, o_dw0, (34 + 32) * 6 + 5)[0])
        for k in range(5):
            tc = 2 * w + k - 2
            if 0 <= tc < 8:
                mg = RX[33 * tc + 32][:, 0]
                dm = dm + _table_row(W, o_dw0, 32 * 6 + k)[0] * mg
                dn = dn + _table_row(W, o_dw0, (34 + 32) * 6 + k)[0] * (lg(mg) - mm)
        xm = RX[33 * 2 * w + 32][:, 0]
        xn = lg(xm) - mm
        part[w, w] += (np.outer(wn[0], np.maximum(dm, 0)) + np.outer(wn[1], xm) + np.outer(wn[2], np.maximum(dn, 0)) +
                       np.outer(wn[3], xn))
    first = np.maximum(part.sum(axis=0), 0)       # [column][16 channels][16 streams]
    R_A16 = 264
    for c in range(4):
        _store16(RX, R_A16 + 4 * c, first[c], relu=False)
    R_Y0, R_Y1, R_Y2, R_Y3, R_Y4, R_Y5, R_Y6 = 0, 16, 48, 64, 80, 88, 104
    R_H0, R_H1, R_H0N, R_H1N = 120, 136, 152, 96
    R8_Y4, R8_Y5, R8_Y6, R_H0M = 80, 0, 32, 0
    RX[R_H0:R_H0 + 32] = hc[:, :128].astype(np.float64).reshape(16, 32, 4).transpose(1, 0, 2)

    def dwq(tab_blk, q, taps):
        d = np.repeat(_table_row(W, tab_blk, q * 6 + 5)[None], 16, 0)
        for k, quad in taps:
            d = d + _table_row(W, tab_blk, q * 6 + k)[None] * quad
        return np.maximum(d, 0)

    def dwfrag(tab_blk, j, taps_of):                # fragment of k-iteration j: lane (n, kq) = channel quad 4 j + kq
        return np.concatenate([dwq(tab_blk, 4 * j + kq, taps_of(4 * j + kq)) for kq in range(4)], axis=0)

    # P3: s0, wave w = column w
    o = sect[0][S["S_S0"]]
    res = {w: vec(o) + _mfma16(W[o + 1], _rows16(RX, R_A16 + 4 * w)) for w in range(4)}
    for w in range(4):
        _store16(RX, R_Y0 + 4 * w, res[w])
    # P4: block 1 (16 -> 32), wave w = column w, two row tiles
    o = sect[0][S["S_L1"]]
    res = {}
    for w in range(4):
        d = dwfrag(o, 0, lambda q: [(k, RX[R_Y0 + 4 * (w + k - 2) + q]) for k in range(5) if 0 <= w + k - 2 < 4])
        y = _rows16(RX, R_Y0 + 4 * w)
        for rt in range(2):
            b = o + 1 + 3 * rt
            res[w, rt] = vec(b) + _mfma16(W[b + 1], d) + _mfma16(W[b + 2], y)
    for (w, rt), acc in res.items():
        _store16(RX, R_Y1 + 8 * w + 4 * rt, acc)
    # P5: s1 on columns 0 and 2; wave w = (column w >> 1, row tile w & 1)
    o = sect[0][S["S_S1"]]
    res = {}
    for w in range(4):
        col, rt = w >> 1, w & 1
        b = o + 3 * rt
        res[w] = vec(b) + sum(_mfma16(W[b + 1 + j], _rows16(RX, R_Y1 + 8 * (2 * col) + 4 * j)) for j in range(2))
    for w in range(4):
        _store16(RX, R_Y2 + 8 * (w >> 1) + 4 * (w & 1), res[w])
    # P6: block 2 (identity residual)
    o = sect[0][S["S_L2"]]
    res = {}
    for w in range(4):
        col, rt = w >> 1, w & 1
        b = o + 1 + 3 * rt
        acc = vec(b)
        for j in range(2):
            acc = acc + _mfma16(W[b + 1 + j], dwfrag(o, j, lambda q: [(k, RX[R_Y2 + 8 * (col + k - 2) + q]) for k in range(5) if 0 <= col + k - 2 < 2]))
        resid = np.concatenate([RX[R_Y2 + 8 * col + 4 * rt + rq].T for rq in range(4)], axis=0)     # [16 ch, 16 streams]
        res[w] = acc + resid
    for w in range(4):
        _store16(RX, R_Y3 + 8 * (w >> 1) + 4 * (w & 1), res[w])
    # P7: s2.  16 kHz: stride 2 -> column 0, waves 0, 1 = row tile; 8 kHz: stride 1, wave w = (column, row tile)
    o = sect[0][S["S_S2"]]
    res = {}
    for w in range(4 if k8 else 2):
        col, rt = (w >> 1, w & 1) if k8 else (0, w)
        b = o + 3 * rt
        res[w] = vec(b) + sum(_mfma16(W[b + 1 + j], _rows16(RX, R_Y3 + 8 * col + 4 * j)) for j in range(2))
    for w, acc in res.items():
        col, rt = (w >> 1, w & 1) if k8 else (0, w)
        _store16(RX, (R8_Y4 + 8 * col if k8 else R_Y4) + 4 * rt, acc)
    # P8: block 3 (32 -> 64), wave w = row tile w; 8 kHz: both columns
    o = sect[0][S["S_L3"]]
    ncol = 2 if k8 else 1
    res = {}
    for w in range(4):
        b = o + 1 + 5 * w
        for col in range(ncol):
            base = R8_Y4 + 8 * col if k8 else R_Y4
            acc = vec(b)
            for j in range(2):
                if k8:
                    oth = 1 - col
                    d = dwfrag(o, j, lambda q: [(2, RX[R8_Y4 + 8 * col + q]), (2 + (oth - col), RX[R8_Y4 + 8 * oth + q])])
                else:
                    d = dwfrag(o, j, lambda q: [(2, RX[R_Y4 + q])])
                acc = acc + _mfma16(W[b + 1 + j], d) + _mfma16(W[b + 3 + j], _rows16(RX, base + 4 * j))
            res[w, col] = acc
    for (w, col), acc in res.items():
        _store16(RX, (R8_Y5 + 16 * col if k8 else R_Y5) + 4 * w, acc)
    # P9: s3 (64 -> 64), wave w = row tile w
    o = sect[0][S["S_S3"]]
    res = {}
    for w in range(4):
        b = o + 5 * w
        for col in range(ncol):
            base = R8_Y5 + 16 * col if k8 else R_Y5
            res[w, col] = vec(b) + sum(_mfma16(W[b + 1 + j], _rows16(RX, base + 4 * j)) for j in range(4))
    for (w, col), acc in res.items():
        _store16(RX, (R8_Y6 + 16 * col if k8 else R_Y6) + 4 * w, acc)
    # LSTMs: wave w = units 16 w .. 16 w + 15, gates i, f, g, o as four 16-row tiles
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    new = hc.astype(np.float64).copy()
    cst = [hc[:, 128:192].astype(np.float64).T.copy(), hc[:, 192:256].astype(np.float64).T.copy()]     # [64 units, 16 streams]
    oh = sect[0][S["S_HEADB"]]
    probs = []
    for step in range(2 if k8 else 1):
        z = np.zeros(16)
        for layer in range(2):
            xin = ((R8_Y6 + 16 * step) if k8 else R_Y6) if layer == 0 else (R_H0N if step == 0 else R_H0M)
            hin = (R_H0 if step == 0 else R_H0N) if layer == 0 else (R_H1 if step == 0 else R_H1N)
            hn_all = np.zeros((64, 16))
            for w in range(4):
                ob = sect[w][S["S_LSTM0" if layer == 0 else "S_LSTM1"]]
                g = [vec(ob + q) for q in range(4)]
                for j in range(8):
                    a = _rows16(RX, (xin if j < 4 else hin) + 4 * (j & 3))
                    for q in range(4):
                        g[q] = g[q] + _mfma16(W[ob + 4 + 4 * j + q], a)
                sl = slice(16 * w, 16 * w + 16)
                cn = sig(g[1]) * cst[layer][sl] + sig(g[0]) * np.tanh(g[2])
                hn = sig(g[3]) * np.tanh(cn)
                cst[layer][sl] = cn
                hn_all[sl] = hn
                new[:, 128 + 64 * layer + 16 * w:128 + 64 * layer + 16 * w + 16] = cn.T
                new[:, 64 * layer + 16 * w:64 * layer + 16 * w + 16] = hn.T
                if layer == 1:
                    z += (_vec16(W[oh + 1 + w])[:, None] * np.maximum(hn, 0)).sum(0)
            dst = (R_H0N if step == 0 else R_H0M) if layer == 0 else R_H1N
            if layer == 0 or step == 0:
                RX[dst:dst + 16] = hn_all.T.reshape(16, 16, 4).transpose(1, 0, 2)
        probs.append(sig(z + W[oh][0, 0]))
    prob = np.mean(probs, axis=0)
    return prob.astype(np.float32), new.astype(np.float32)

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
    lib = _ffi.lib()
    n = C.c_size_t()
    sect = (C.c_uint32 * 32)()
    rc = lib.vad_debug_pack_weights(version, blob, len(blob), None, 0, C.byref(n), sect)
    if rc != 0:
        raise RuntimeError(lib.vad_last_create_error().decode())
    out = np.empty(n.value, np.float32)
    rc = lib.vad_debug_pack_weights(version, blob, len(blob), out.ctypes.data_as(C.POINTER(C.c_float)), out.size,
                                    C.byref(n), sect)
    assert rc == 0
    return out.reshape(-1, 64, 4), np.array(sect, dtype=np.int64).reshape(4, 8)


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


def v5_step(W, sect, x, hc, gate=0.01):
    """x [32,512] f32, hc [32,256] -> (prob [32], new hc [32,256]).  float64 contractions.
    Mirrors silero_v5.hip: folded loader, one activation region RX (row map in vad_layout.h)."""
    x = x.astype(np.float64)
    if gate is not None and gate >= 0:
        x = np.where(np.abs(x) > gate, x, 0.0)
    RX = np.zeros((194, 32, 4))
    RH = np.zeros((32, 32, 4))
    RE = RX[98:]
    # loader: fold every column (c = 0..2): u[n] = x[n] + x[256-n], v[n] = x[n] - x[256-n], n = 1..128
    for c in range(3):
        col = x[:, 128 * c:128 * c + 256] if c < 2 else np.concatenate([x[:, 256:512]], axis=1)
        n = np.arange(1, 129)
        mir = np.where(n < 128, 256 - n, 0)
        xm = np.where(n[None, :] < 128, col[:, mir], 0.0)
        u = col[:, n] + xm
        v = np.where(n[None, :] < 128, col[:, n] - xm, 0.0)
        RX[64 * c:64 * c + 32] = u.reshape(32, 32, 4).transpose(1, 0, 2)
        RX[64 * c + 32:64 * c + 64] = v.reshape(32, 32, 4).transpose(1, 0, 2)
    RH[:] = hc[:, :128].astype(np.float64).reshape(32, 32, 4).transpose(1, 0, 2)
    c_prev = hc[:, 128:].astype(np.float64)
    # bin 128 on the VALU
    nq = W[sect[0][S_NYQ]].reshape(-1)[:128].astype(np.float64)
    nyq = np.zeros((3, 32))
    for c in range(3):
        u = RX[64 * c:64 * c + 32].transpose(1, 0, 2).reshape(32, 128)
        nyq[c] = np.abs(u @ nq)
    # STFT
    mags = {}
    for w in range(4):
        ws = sect[w][S_STFT]
        are = [np.zeros((32, 32)) for _ in range(3)]
        aim = [np.zeros((32, 32)) for _ in range(3)]
        for j in range(16):
            wre, wim = W[ws + 2 * j], W[ws + 2 * j + 1]
            for c in range(3):
                are[c] += _mfma4(wre, _rows(RX, 64 * c + 2 * j, 64 * c + 2 * j + 1))
                aim[c] += _mfma4(wim, _rows(RX, 64 * c + 32 + 2 * j, 64 * c + 32 + 2 * j + 1))
        mags[w] = [np.sqrt(are[c] ** 2 + aim[c] ** 2) for c in range(3)]
    for w in range(4):
        for c in range(3):
            _store_tile(RX, c * 32 + 8 * w, mags[w][c], relu=False)
    RX[96] = np.concatenate([nyq.T, np.zeros((32, 1))], axis=1)
    RX[97] = 0
    # enc0
    E0 = {}
    for w in range(4):
        ws = sect[w][S_ENC0]
        bias = _vec(W[ws:ws + 4])
        acc = [np.repeat(bias[:, None], 32, 1) for _ in range(3)]
        ws += 4
        for j in range(16):
            wt = [W[ws + 3 * j + t] for t in range(3)]
            a = [_rows(RX, 32 * c + 2 * j, 32 * c + 2 * j + 1) for c in range(3)]
            acc[0] += _mfma4(wt[1], a[0]) + _mfma4(wt[2], a[1])
            acc[1] += _mfma4(wt[0], a[0]) + _mfma4(wt[1], a[1]) + _mfma4(wt[2], a[2])
            acc[2] += _mfma4(wt[0], a[1]) + _mfma4(wt[1], a[2])
        an = _rows(RX, 96, 97)
        for c in range(3):
            acc[c] += _mfma4(W[ws + 48 + c], an)
        E0[w] = acc
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
    E2 = {}
    for w in range(2):
        ws = sect[w][S_ENC2]
        acc = np.repeat(_vec(W[ws:ws + 4])[:, None], 32, 1)
        ws += 4
        for it in range(16):
            ti, j = it >> 3, it & 7
            r = ti * 16 + 2 * j
            acc += _mfma4(W[ws + it], _rows(RX, r, r + 1))
        E2[w] = acc
    for w in range(2):
        _store_tile(RE, 8 * w, E2[w])
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
        g = [np.repeat(_vec(W[ws + 4 * q:ws + 4 * q + 4])[:, None], 32, 1) for q in range(4)]
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

// Fused Silero-VAD V5 (16 kHz) step kernel for MI355X / gfx950.
//
// Replaces `session.run` on silero_vad_v5.onnx's 16 kHz branch for a batch of independent
// streams (reference call site: /root/reference/src/real_time_vad/core/silero_model.py:433;
// dataflow: SURVEY.md §8 a7) together with the pre-steps the reference does in Python per
// frame: the denoise gate (utils/audio.py:117-118), the int16->float scaling
// (websocket_service/server/vad_websocket_server.py:341) and the hysteresis state machine
// (core/silero_model.py:790-949).
//
// One workgroup (4 waves) carries a tile of 32 streams through the WHOLE network:
//   load+gate -> STFT (windowed DFT as MFMA GEMM) -> |.| -> enc0..enc3 (+ReLU) -> LSTM cell
//   -> head -> sigmoid -> state machine.
// Activations never leave the CU (LDS quads, see vad_layout.h); weights stream from L2 into
// VGPRs in packed per-wave order; all contractions are v_mfma_f32_32x32x2_f32 (exact fp32).
// T > 1 frames per stream are processed in-kernel with h in LDS and c in registers.
#include <hip/hip_runtime.h>
#include "vad_layout.h"
#include "sm_device.h"

using namespace vadk;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x16 mfma4(f32x4 w, f32x4 a, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a.w, acc, 0, 0, 0);
    return acc;
}

// accumulator initialised from 4 "lane-expanded" bias blocks (regs 4g..4g+3 <- block g)
__device__ __forceinline__ f32x16 acc_from(const f32x4 *ws) {
    f32x4 b0 = ws[0], b1 = ws[BLK_F4], b2 = ws[2 * BLK_F4], b3 = ws[3 * BLK_F4];
    f32x16 a;
    a.s0 = b0.x; a.s1 = b0.y; a.s2 = b0.z; a.s3 = b0.w;
    a.s4 = b1.x; a.s5 = b1.y; a.s6 = b1.z; a.s7 = b1.w;
    a.s8 = b2.x; a.s9 = b2.y; a.sa = b2.z; a.sb = b2.w;
    a.sc = b3.x; a.sd = b3.y; a.se = b3.z; a.sf = b3.w;
    return a;
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

__device__ __forceinline__ f32x4 quad_of(const f32x16 &a, int g) {
    switch (g) {
        case 0: return f32x4{a.s0, a.s1, a.s2, a.s3};
        case 1: return f32x4{a.s4, a.s5, a.s6, a.s7};
        case 2: return f32x4{a.s8, a.s9, a.sa, a.sb};
        default: return f32x4{a.sc, a.sd, a.se, a.sf};
    }
}

// write a 32-channel output tile (relu'd) as 8 quad rows starting at row `row0`;
// lane (m,h) owns quads row0 + 2g + h
__device__ __forceinline__ void store_tile_relu(f32x4 *region, int row0, int m, int h, const f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) region[(row0 + 2 * g + h) * QS + m] = relu4(quad_of(acc, g));
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ f32x4 gate4(f32x4 v, float thr) {
    // utils/audio.py:117-118: np.where(np.abs(x) > thr, x, 0.0); thr < 0 disables the gate
    if (thr >= 0.f) {
        v.x = fabsf(v.x) > thr ? v.x : 0.f;
        v.y = fabsf(v.y) > thr ? v.y : 0.f;
        v.z = fabsf(v.z) > thr ? v.z : 0.f;
        v.w = fabsf(v.w) > thr ? v.w : 0.f;
    }
    return v;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(NTHREADS, 1) silero_v5_step(const StepParams P) {
    using namespace vadk::v5;
    __shared__ f32x4 lds[LDS_F4];
    f32x4 *const RA = lds;                       // x / enc0 / enc2
    f32x4 *const RB = lds + ROWS_A * QS;         // mag / enc1 / enc3
    f32x4 *const RH = RB + ROWS_B * QS;          // h
    float *const headp = reinterpret_cast<float *>(RH + ROWS_H * QS);   // [4][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;
    const int h = lane >> 5;
    const int tile0 = blockIdx.x * MT;
    const int gf = tile0 + m;                     // this lane's stream index within the call
    const bool live = gf < P.n;
    const int slot = live ? (P.slots ? P.slots[gf] : gf) : 0;
    const f32x4 *const wbase = reinterpret_cast<const f32x4 *>(P.wstream) + lane;
    const int T = P.T;

    // ---- prologue: h_{t-1} -> LDS quads, c_{t-1} -> registers (this lane's 16 units) -------
    {
        const int fm = tid & 31, part = tid >> 5;
        const int g2 = tile0 + fm;
        const int s2 = g2 < P.n ? (P.slots ? P.slots[g2] : g2) : -1;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = part * 4 + qq;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s2 >= 0) v = reinterpret_cast<const f32x4 *>(P.state + (size_t)s2 * 256)[q];
            RH[q * QS + fm] = v;
        }
    }
    f32x16 cst;   // c state of units 32w + 8g + 4h + i  (reg 4g+i)
    {
        f32x4 c4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            c4[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live) c4[g] = *reinterpret_cast<const f32x4 *>(P.state + (size_t)slot * 256 + 128 + 32 * w + 8 * g + 4 * h);
        }
        cst.s0 = c4[0].x; cst.s1 = c4[0].y; cst.s2 = c4[0].z; cst.s3 = c4[0].w;
        cst.s4 = c4[1].x; cst.s5 = c4[1].y; cst.s6 = c4[1].z; cst.s7 = c4[1].w;
        cst.s8 = c4[2].x; cst.s9 = c4[2].y; cst.sa = c4[2].z; cst.sb = c4[2].w;
        cst.sc = c4[3].x; cst.sd = c4[3].y; cst.se = c4[3].z; cst.sf = c4[3].w;
    }
    f32x16 hst = cst;  // overwritten before use; keeps the h' of the last frame for the HBM write-back
    SmSlot sm;
    int seg_last = 0;
    const bool sm_thread = (tid < MT) && (tile0 + tid < P.n);
    int sm_slot = 0;
    if (sm_thread) {
        sm_slot = P.slots ? P.slots[tile0 + tid] : tile0 + tid;
        sm = P.sm[sm_slot];
    }

    for (int t = 0; t < T; ++t) {
        // ---- load + convert + gate one frame per stream into region A (transposed to quads) ----
        {
            const float thr = P.thresh;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int idx = it * NTHREADS + tid;
                const int fm = idx >> 7, q = idx & 127;
                const int g2 = tile0 + fm;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g2 < P.n) {
                    const size_t fo = ((size_t)g2 * T + t) * 128 + q;   // in units of 4 samples
                    if (P.fmt == 0) {
                        v = reinterpret_cast<const f32x4 *>(P.frames)[fo];
                    } else {
                        const i16x4 s = reinterpret_cast<const i16x4 *>(P.frames)[fo];
                        const float sc = P.fmt == 1 ? 32767.0f : 32768.0f;
                        // the reference divides (np.int16 -> float32 / 32767.0), keep a true division
                        v = f32x4{(float)s.x / sc, (float)s.y / sc, (float)s.z / sc, (float)s.w / sc};
                    }
                    v = gate4(v, thr);
                }
                RA[q * QS + fm] = v;
            }
        }
        __syncthreads();   // (1) x and h visible

        // ---- STFT: wave w computes bins 32w..32w+31 (re and im) for the 3 columns ----------
        {
            const f32x4 *ws = wbase + (size_t)P.sect[w][S_STFT] * BLK_F4;
            const f32x4 *wn = wbase + (size_t)P.sect[w][S_NYQ] * BLK_F4;
            f32x16 are[3], aim[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                are[c] = (f32x16)(0.f);
                aim[c] = (f32x16)(0.f);
            }
            float nre[3] = {0.f, 0.f, 0.f}, nim[3] = {0.f, 0.f, 0.f};
            f32x4 wre = ws[0], wim = ws[BLK_F4];
            f32x4 x0 = RA[(0 + h) * QS + m], x1 = RA[(32 + h) * QS + m], x2 = RA[(64 + h) * QS + m];
            f32x4 wnr = f32x4{0.f, 0.f, 0.f, 0.f}, wni = wnr;
            if (w == 3) { wnr = wn[0]; wni = wn[BLK_F4]; }
            for (int j = 0; j < 32; ++j) {
                const int jn = j < 31 ? j + 1 : 31;
                const f32x4 nwre = ws[(2 * jn) * BLK_F4], nwim = ws[(2 * jn + 1) * BLK_F4];
                const f32x4 nx0 = RA[(2 * jn + h) * QS + m];
                const f32x4 nx1 = RA[(32 + 2 * jn + h) * QS + m];
                const f32x4 nx2 = RA[(64 + 2 * jn + h) * QS + m];
                f32x4 nwnr = wnr, nwni = wni;
                if (w == 3) { nwnr = wn[(2 * jn) * BLK_F4]; nwni = wn[(2 * jn + 1) * BLK_F4]; }
                are[0] = mfma4(wre, x0, are[0]);
                are[1] = mfma4(wre, x1, are[1]);
                are[2] = mfma4(wre, x2, are[2]);
                aim[0] = mfma4(wim, x0, aim[0]);
                aim[1] = mfma4(wim, x1, aim[1]);
                aim[2] = mfma4(wim, x2, aim[2]);
                if (w == 3) {   // bin 128 (Nyquist) on the VALU, hidden under the MFMAs
                    nre[0] += wnr.x * x0.x + wnr.y * x0.y + wnr.z * x0.z + wnr.w * x0.w;
                    nre[1] += wnr.x * x1.x + wnr.y * x1.y + wnr.z * x1.z + wnr.w * x1.w;
                    nre[2] += wnr.x * x2.x + wnr.y * x2.y + wnr.z * x2.z + wnr.w * x2.w;
                    nim[0] += wni.x * x0.x + wni.y * x0.y + wni.z * x0.z + wni.w * x0.w;
                    nim[1] += wni.x * x1.x + wni.y * x1.y + wni.z * x1.z + wni.w * x1.w;
                    nim[2] += wni.x * x2.x + wni.y * x2.y + wni.z * x2.z + wni.w * x2.w;
                }
                wre = nwre; wim = nwim; x0 = nx0; x1 = nx1; x2 = nx2; wnr = nwnr; wni = nwni;
            }
            // magnitude -> region B rows c*32 + 8w + 2g + h
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 r = quad_of(are[c], g), i = quad_of(aim[c], g);
                    RB[(c * 32 + 8 * w + 2 * g + h) * QS + m] =
                        f32x4{sqrtf(r.x * r.x + i.x * i.x), sqrtf(r.y * r.y + i.y * i.y),
                              sqrtf(r.z * r.z + i.z * i.z), sqrtf(r.w * r.w + i.w * i.w)};
                }
            }
            if (w == 3) {
                float mg[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float r = nre[c] + __shfl_xor(nre[c], 32);
                    const float i = nim[c] + __shfl_xor(nim[c], 32);
                    mg[c] = sqrtf(r * r + i * i);
                }
                // row 96: (|X128| of column 0,1,2, 0) ; row 97: zeros (pairs with row 96 in the MFMA k-step)
                RB[(96 + h) * QS + m] = h == 0 ? f32x4{mg[0], mg[1], mg[2], 0.f} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();   // (2) mag complete; region A free

        // ---- enc0: 129 -> 128 ch, k3 s1 p1, 3 -> 3 columns; wave w: channels 32w.. ---------
        {
            const f32x4 *ws = wbase + (size_t)P.sect[w][S_ENC0] * BLK_F4;
            f32x16 acc[3];
            acc[0] = acc_from(ws);
            acc[1] = acc[0];
            acc[2] = acc[0];
            ws += 4 * BLK_F4;
            f32x4 w0 = ws[0], w1 = ws[BLK_F4], w2 = ws[2 * BLK_F4];
            f32x4 a0 = RB[(0 + h) * QS + m], a1 = RB[(32 + h) * QS + m], a2 = RB[(64 + h) * QS + m];
            for (int j = 0; j < 16; ++j) {
                const int jn = j < 15 ? j + 1 : 15;
                const f32x4 nw0 = ws[(3 * jn) * BLK_F4], nw1 = ws[(3 * jn + 1) * BLK_F4], nw2 = ws[(3 * jn + 2) * BLK_F4];
                const f32x4 na0 = RB[(2 * jn + h) * QS + m];
                const f32x4 na1 = RB[(32 + 2 * jn + h) * QS + m];
                const f32x4 na2 = RB[(64 + 2 * jn + h) * QS + m];
                // out[c] += W[tap] * in[c + tap - 1]
                acc[0] = mfma4(w1, a0, acc[0]);
                acc[1] = mfma4(w0, a0, acc[1]);
                acc[2] = mfma4(w0, a1, acc[2]);
                acc[0] = mfma4(w2, a1, acc[0]);
                acc[1] = mfma4(w1, a1, acc[1]);
                acc[2] = mfma4(w1, a2, acc[2]);
                acc[1] = mfma4(w2, a2, acc[1]);
                w0 = nw0; w1 = nw1; w2 = nw2; a0 = na0; a1 = na1; a2 = na2;
            }
            {   // input channel 128 (Nyquist bin): one k-iteration against per-column weight blocks
                const f32x4 an = RB[(96 + h) * QS + m];
                acc[0] = mfma4(ws[48 * BLK_F4], an, acc[0]);
                acc[1] = mfma4(ws[49 * BLK_F4], an, acc[1]);
                acc[2] = mfma4(ws[50 * BLK_F4], an, acc[2]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) store_tile_relu(RA, c * 32 + 8 * w, m, h, acc[c]);
        }
        __syncthreads();   // (3) enc0 out in A; region B free

        // ---- enc1: 128 -> 64 ch, k3 s2 p1, 3 -> 2 columns; wave w: n-tile w&1, column w>>1 ----
        {
            const int nt = w & 1, tp = w >> 1;
            const f32x4 *ws = wbase + (size_t)P.sect[w][S_ENC1] * BLK_F4;
            f32x16 acc = acc_from(ws);
            ws += 4 * BLK_F4;
            // valid taps: tp=0 -> taps 1,2 on input columns 0,1 ; tp=1 -> taps 0,1 on columns 1,2
#pragma unroll 8
            for (int it = 0; it < 32; ++it) {
                const int ti = it >> 4, j = it & 15;
                const f32x4 wv = ws[it * BLK_F4];
                const f32x4 av = RA[((tp + ti) * 32 + 2 * j + h) * QS + m];
                acc = mfma4(wv, av, acc);
            }
            store_tile_relu(RB, tp * 16 + 8 * nt, m, h, acc);
        }
        __syncthreads();   // (4) enc1 out in B; region A free

        // ---- enc2: 64 -> 64 ch, k3 s2 p1, 2 -> 1 column (taps 1,2 on columns 0,1); waves 0,1 ----
        if (w < 2) {
            const f32x4 *ws = wbase + (size_t)P.sect[w][S_ENC2] * BLK_F4;
            f32x16 acc = acc_from(ws);
            ws += 4 * BLK_F4;
#pragma unroll 8
            for (int it = 0; it < 16; ++it) {
                const int ti = it >> 3, j = it & 7;
                const f32x4 wv = ws[it * BLK_F4];
                const f32x4 av = RB[(ti * 16 + 2 * j + h) * QS + m];
                acc = mfma4(wv, av, acc);
            }
            store_tile_relu(RA, 8 * w, m, h, acc);
        }
        __syncthreads();   // (5) enc2 out in A

        // ---- enc3: 64 -> 128 ch, k3 s1 p1 on a single column: centre tap only ------------------
        {
            const f32x4 *ws = wbase + (size_t)P.sect[w][S_ENC3] * BLK_F4;
            f32x16 acc = acc_from(ws);
            ws += 4 * BLK_F4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 wv = ws[j * BLK_F4];
                const f32x4 av = RA[(2 * j + h) * QS + m];
                acc = mfma4(wv, av, acc);
            }
            store_tile_relu(RB, 8 * w, m, h, acc);
        }
        __syncthreads();   // (6) enc3 out (LSTM input x) in B rows 0..31

        // ---- LSTM cell: wave w owns hidden units 32w..32w+31, all four gates -------------------
        f32x16 gi, gfo, gg, go;
        {
            const f32x4 *ws = wbase + (size_t)P.sect[w][S_LSTM] * BLK_F4;
            gi = acc_from(ws);
            gfo = acc_from(ws + 4 * BLK_F4);
            gg = acc_from(ws + 8 * BLK_F4);
            go = acc_from(ws + 12 * BLK_F4);
            ws += 16 * BLK_F4;
            const f32x4 *src = RB;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                f32x4 wi = ws[0], wf = ws[BLK_F4], wg = ws[2 * BLK_F4], wo = ws[3 * BLK_F4];
                f32x4 av = src[h * QS + m];
                for (int j = 0; j < 16; ++j) {
                    const int jn = j < 15 ? j + 1 : 15;
                    const f32x4 nwi = ws[(4 * jn) * BLK_F4], nwf = ws[(4 * jn + 1) * BLK_F4];
                    const f32x4 nwg = ws[(4 * jn + 2) * BLK_F4], nwo = ws[(4 * jn + 3) * BLK_F4];
                    const f32x4 nav = src[(2 * jn + h) * QS + m];
                    gi = mfma4(wi, av, gi);
                    gfo = mfma4(wf, av, gfo);
                    gg = mfma4(wg, av, gg);
                    go = mfma4(wo, av, go);
                    wi = nwi; wf = nwf; wg = nwg; wo = nwo; av = nav;
                }
                ws += 64 * BLK_F4;
                src = RH;
            }
            // ws now points at the 4 head-weight blocks
            __syncthreads();   // (7) every wave is done reading h_{t-1}; region A free for the next frame
            float part = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 i4 = quad_of(gi, g), f4 = quad_of(gfo, g), g4 = quad_of(gg, g), o4 = quad_of(go, g);
                const f32x4 c4 = quad_of(cst, g);
                const f32x4 hw = ws[g * BLK_F4];
                f32x4 cn, hn;
#define CELL(k)                                                             \
    cn.k = sigmoidf_(f4.k) * c4.k + sigmoidf_(i4.k) * tanhf(g4.k);        \
    hn.k = sigmoidf_(o4.k) * tanhf(cn.k);                                 \
    part += hw.k * fmaxf(hn.k, 0.f);
                CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
                RH[(8 * w + 2 * g + h) * QS + m] = hn;
                switch (g) {
                    case 0: cst.s0 = cn.x; cst.s1 = cn.y; cst.s2 = cn.z; cst.s3 = cn.w;
                            hst.s0 = hn.x; hst.s1 = hn.y; hst.s2 = hn.z; hst.s3 = hn.w; break;
                    case 1: cst.s4 = cn.x; cst.s5 = cn.y; cst.s6 = cn.z; cst.s7 = cn.w;
                            hst.s4 = hn.x; hst.s5 = hn.y; hst.s6 = hn.z; hst.s7 = hn.w; break;
                    case 2: cst.s8 = cn.x; cst.s9 = cn.y; cst.sa = cn.z; cst.sb = cn.w;
                            hst.s8 = hn.x; hst.s9 = hn.y; hst.sa = hn.z; hst.sb = hn.w; break;
                    default: cst.sc = cn.x; cst.sd = cn.y; cst.se = cn.z; cst.sf = cn.w;
                             hst.sc = hn.x; hst.sd = hn.y; hst.se = hn.z; hst.sf = hn.w; break;
                }
            }
            part += __shfl_xor(part, 32);
            if (h == 0) headp[w * 32 + m] = part;
        }
        __syncthreads();   // (8) head partials + new h visible

        // ---- head: p = sigmoid(b + sum_j w_j relu(h'_j)); then the state machine ---------------
        if (tid < MT) {
            const float hb = P.wstream[(size_t)P.sect[0][S_HEADB] * BLK_FLOATS];
            const float z = hb + ((headp[tid] + headp[32 + tid]) + (headp[64 + tid] + headp[96 + tid]));
            const float p = sigmoidf_(z);
            if (sm_thread) {
                P.probs[(size_t)(tile0 + tid) * T + t] = p;
                int seg = 0;
                const int ev = sm_step(sm, p, &seg);
                if (ev & 2) seg_last = seg;
                if (P.events) P.events[(size_t)(tile0 + tid) * T + t] = (uint8_t)ev;
            }
        }
        // no barrier needed here: the next frame's loader only writes region A (free since (7)),
        // headp is rewritten only after barriers (1)..(7) of the next frame.
    }

    // ---- epilogue: state write-back ---------------------------------------------------------
    if (live) {
        float *st = P.state + (size_t)slot * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<f32x4 *>(st + 32 * w + 8 * g + 4 * h) = quad_of(hst, g);
            *reinterpret_cast<f32x4 *>(st + 128 + 32 * w + 8 * g + 4 * h) = quad_of(cst, g);
        }
    }
    if (sm_thread) {
        P.sm[sm_slot] = sm;
        if (P.seg_frames) P.seg_frames[tile0 + tid] = seg_last;
    }
}

// host-callable launcher (engine.cpp is plain C++ and never sees <<<>>>)
extern "C" hipError_t vadk_launch_silero_v5(const vadk::StepParams *p, hipStream_t stream) {
    const int tiles = (p->n + vadk::MT - 1) / vadk::MT;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(silero_v5_step, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p);
    return hipGetLastError();
}

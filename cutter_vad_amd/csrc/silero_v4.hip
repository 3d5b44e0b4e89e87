// Fused Silero-VAD V4 (16 kHz) step kernels for MI355X / gfx950.
//
// Replaces `session.run` on silero_vad.onnx's 16 kHz branch for a batch of independent streams
// (reference call site: /root/reference/src/real_time_vad/core/silero_model.py:433 with the feeds
// of :494-499; dataflow: SURVEY.md §8 a8), plus the per-frame pre-steps and the state machine, as
// in silero_v5.hip.  Two launches per frame (vad_layout.h explains why):
//
//   silero_v4_stft : load + gate + reflect-pad(96,96) + fold -> 8-column windowed DFT (MFMA) -> |.|
//                    -> global scratch [tile][8][33 quads][32 streams]
//   silero_v4_tail : log(1 + |X| 2^20), adaptive normalisation, first layer (dw k5 + pw + proj),
//                    1x1 stride convs, 3 separable blocks, LSTM(64) x 2, head, state machine
//
// Same conventions as the V5 kernel: 32 streams per workgroup, weights on the MFMA A operand,
// activations as LDS quads on the B operand, packed per-wave weight streams read through one
// buffer descriptor, fenced software pipelining.
#include <hip/hip_runtime.h>
#include "vad_layout.h"
#include "sm_device.h"
#include "vadk_device.h"

using namespace vadk;
using namespace vadk::dev;

namespace {

// xp = reflect-pad(x, 96, 96) (numpy 'reflect'), as aligned quads: xp[4Q .. 4Q+3], Q in 0..175, from
// the raw frame xs[0..127] (quads of x[512]):  xp[i] = x[96-i] (i<96), x[i-96] (i<608), x[1118-i] (else)
__device__ __forceinline__ f32x4 xp_quad(const f32x4 *xs, int Q) {
    if (Q >= 24 && Q < 152) return xs[Q - 24];
    if (Q < 24) {
        const f32x4 lo = xs[24 - Q], hi = xs[23 - Q];          // x[96-4Q] | x[95-4Q], x[94-4Q], x[93-4Q]
        return f32x4{lo.x, hi.w, hi.z, hi.y};
    }
    const f32x4 p = xs[279 - Q], pm = xs[278 - Q];              // x[1118-4Q] = quad (279-Q) comp 2
    return f32x4{p.z, p.y, p.x, pm.w};
}

}  // namespace

// =====================================================================================================
//  launch 1: STFT magnitudes
// =====================================================================================================
extern "C" __global__ void __launch_bounds__(NTHREADS, 1) silero_v4_stft(const StepParams P, const int tframe) {
    using namespace vadk::v4;
    __shared__ f32x4 lds[K1_LDS_F4];
    f32x4 *const XS = lds;                          // raw frame, [32 streams][128 quads]
    f32x4 *const UV = lds + K1_XS_F4;               // u/v of the two columns in flight: rows 64c' + q | 64c' + 32 + q
    float *const nyqv = reinterpret_cast<float *>(UV + K1_UV_ROWS * QS);   // [2][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int hq = h * QS + m;
    const int tile0 = blockIdx.x * MT;
    const int T = P.T;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const int lane16 = lane * 16;
    const int o_stft = (int)P.sect[w][S_STFT], o_nyq = (int)P.sect[w][S_NYQ];
    f32x4 *const scratch = reinterpret_cast<f32x4 *>(P.scratch) + (size_t)blockIdx.x * SCRATCH_F4_PER_TILE;

    // ---- raw frame -> LDS (gate + int16 scaling fused), lanes run along the frame ---------------
    {
        const float thr = P.thresh;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            if (P.fmt == 0) {
                f32x4 xv[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int idx = (half * 8 + it) * NTHREADS + tid;
                    const int g2 = tile0 + (idx >> 7);
                    xv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (g2 < P.n) xv[it] = reinterpret_cast<const f32x4 *>(P.frames)[((size_t)g2 * T + tframe) * 128 + (idx & 127)];
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) XS[(half * 8 + it) * NTHREADS + tid] = gate4(xv[it], thr);
            } else {
                const float sc = P.fmt == 1 ? 32767.0f : 32768.0f;
#pragma unroll 1
                for (int it = 0; it < 8; ++it) {
                    const int idx = (half * 8 + it) * NTHREADS + tid;
                    const int g2 = tile0 + (idx >> 7);
                    i16x4 s = i16x4{0, 0, 0, 0};
                    if (g2 < P.n) s = reinterpret_cast<const i16x4 *>(P.frames)[((size_t)g2 * T + tframe) * 128 + (idx & 127)];
                    XS[idx] = gate4(f32x4{(float)s.x / sc, (float)s.y / sc, (float)s.z / sc, (float)s.w / sc}, thr);
                }
            }
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int grp = 0; grp < 4; ++grp) {              // STFT columns 2 grp, 2 grp + 1 (hop 64 on the padded frame)
        int ws = o_stft, wn = o_nyq;
        asm volatile("" : "+s"(ws), "+s"(wn));
        // ---- fold: u[n] = xp[64t+n] + xp[64t+256-n], v = difference, n = 4q+1..4q+4; n = 128 is its own mirror
        {
            const int q = tid & 31;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int o = rr * 8 + (tid >> 5);       // (column c', stream): o = c' * 32 + stream
                const int cp = o >> 5, ms = o & 31;
                const int Q0 = 16 * (2 * grp + cp);      // first xp quad of the column
                const f32x4 *xs = XS + ms * 128;
                const f32x4 a = xp_quad(xs, Q0 + q), b2 = xp_quad(xs, Q0 + q + 1), d = xp_quad(xs, Q0 + 63 - q);
                f32x4 u = f32x4{a.y + d.w, a.z + d.z, a.w + d.y, b2.x + d.x};
                f32x4 v = f32x4{a.y - d.w, a.z - d.z, a.w - d.y, b2.x - d.x};
                if (q == 31) { u.w = b2.x; v.w = 0.f; }
                UV[(64 * cp + q) * QS + ms] = u;
                UV[(64 * cp + 32 + q) * QS + ms] = v;
            }
        }
        f32x4 Are = ldw(wrs, lane16, ws), Aim = ldw(wrs, lane16, ws + 1);
        SB();
        __syncthreads();
        // ---- bin 128 on the VALU: 64 (column, stream) pairs, 4 lanes each
        {
            const int pair = tid >> 2, part = tid & 3;
            const int cp = pair >> 5, ms = pair & 31;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int qq = part * 8 + i;
                const f32x4 cf = ldw(wrs, qq * 16, wn);
                const f32x4 uu = UV[(64 * cp + qq) * QS + ms];
                a += cf.x * uu.x + cf.y * uu.y + cf.z * uu.z + cf.w * uu.w;
            }
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            if (part == 0) nyqv[cp * 32 + ms] = fabsf(a);
        }
        // ---- MFMA: wave w = bins 32w..32w+31, re on u / im on v, two columns
        f32x16 are[2], aim[2];
        are[0] = are[1] = aim[0] = aim[1] = (f32x16)(0.f);
        {
            f32x4 Au0 = UV[0 * QS + hq], Au1 = UV[64 * QS + hq], Av0 = UV[32 * QS + hq], Av1 = UV[96 * QS + hq];
            f32x4 Bre, Bim, Bu0, Bu1, Bv0, Bv1;
#define K1_LD(S, jj)                                                                        \
    S##re = ldw(wrs, lane16, ws + 2 * (jj)); S##im = ldw(wrs, lane16, ws + 2 * (jj) + 1);   \
    S##u0 = UV[(2 * (jj)) * QS + hq]; S##u1 = UV[(64 + 2 * (jj)) * QS + hq];                \
    S##v0 = UV[(32 + 2 * (jj)) * QS + hq]; S##v1 = UV[(96 + 2 * (jj)) * QS + hq];
#define K1_MMA(S)                                                                           \
    are[0] = mfma4(S##re, S##u0, are[0]); are[1] = mfma4(S##re, S##u1, are[1]);             \
    aim[0] = mfma4(S##im, S##v0, aim[0]); aim[1] = mfma4(S##im, S##v1, aim[1]);
            for (int j = 0; j < 16; j += 2) {
                K1_LD(B, j + 1) SB();
                K1_MMA(A) SB();
                const int jn = j + 2 < 16 ? j + 2 : 14;
                K1_LD(A, jn) SB();
                K1_MMA(B) SB();
            }
#undef K1_LD
#undef K1_MMA
        }
        // ---- magnitudes -> global scratch, row (33 t + quad), 32 float4 per row
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int tcol = 2 * grp + cp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 r = quad_of(are[cp], g), i = quad_of(aim[cp], g);
                scratch[(size_t)(MAG_Q * tcol + 8 * w + 2 * g + h) * 32 + m] =
                    f32x4{mag_(r.x, i.x), mag_(r.y, i.y), mag_(r.z, i.z), mag_(r.w, i.w)};
            }
        }
        __syncthreads();       // nyqv complete; every wave done with UV before the next fold overwrites it
        if (tid < 64) scratch[(size_t)(MAG_Q * (2 * grp + h) + 32) * 32 + m] = f32x4{nyqv[h * 32 + m], 0.f, 0.f, 0.f};
    }
}

// =====================================================================================================
//  launch 2: everything after the STFT
// =====================================================================================================
namespace {

// one "thin" MFMA group: 4 k-iterations = 4 weight blocks x 4 activation quads -> 16 MFMAs on one accumulator
#define TG_MMA(acc, w0, w1, w2, w3, a0, a1, a2, a3) \
    acc = mfma4(w0, a0, acc); acc = mfma4(w1, a1, acc); acc = mfma4(w2, a2, acc); acc = mfma4(w3, a3, acc);

__device__ __forceinline__ f32x4 ldt(__amdgpu_buffer_rsrc_t rs, int row, int blk) {   // table row (float4) of a VALU table
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, row * 16, blk * 1024, 0));
}

__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) {
    return f32x4{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}

__device__ __forceinline__ float log1p20(float mag) {   // log(1 + mag * 2^20): Mul, Add, Log of the graph
    return __builtin_amdgcn_logf(1.0f + mag * 1048576.0f) * 0.69314718055994531f;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(NTHREADS, 1) silero_v4_tail(const StepParams P, const int tframe) {
    using namespace vadk::v4;
    __shared__ f32x4 lds[K2_LDS_F4];
    f32x4 *const RX = lds;
    float *const misc = reinterpret_cast<float *>(lds + K2_ROWS * QS);
    float *const mmv = misc;                 // [32]  mean_mean per stream
    float *const colmean = misc + 32;        // [8][32]
    float *const headp = misc + 32 + 256;    // [2][32]
    float *const gpart = reinterpret_cast<float *>(RX + R_GP * QS);   // [2][64 regs][64 lanes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int hq = h * QS + m;
    const int tile0 = blockIdx.x * MT;
    const int gf = tile0 + m;
    const bool live = gf < P.n;
    const int slot = live ? (P.slots ? P.slots[gf] : gf) : 0;
    const int T = P.T;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const int lane16 = lane * 16;
#define WL(blk) ldw(wrs, lane16, (blk))
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *const scratch = reinterpret_cast<const f32x4 *>(P.scratch) + (size_t)blockIdx.x * SCRATCH_F4_PER_TILE;

    // ---- P0: magnitudes scratch -> LDS rows (33 t + q) ------------------------------------------------
#pragma unroll 3
    for (int it = 0; it < 33; ++it) {
        const int idx = it * NTHREADS + tid;          // 264 rows x 32
        RX[(idx >> 5) * QS + (idx & 31)] = scratch[idx];
    }
    __syncthreads();

    // ---- P1: adaptive normalisation scalar: mean over bins per column, reflect-padded 7-tap smoothing, mean over columns
    {
        const int ms = tid & 31, tc = tid >> 5;       // one (column, stream) per thread
        float s = 0.f;
#pragma unroll 4
        for (int q = 0; q < 32; ++q) {
            const f32x4 v = RX[(MAG_Q * tc + q) * QS + ms];
            s += (log1p20(v.x) + log1p20(v.y)) + (log1p20(v.z) + log1p20(v.w));
        }
        s += log1p20(RX[(MAG_Q * tc + 32) * QS + ms].x);
        colmean[tc * 32 + ms] = s * (1.0f / 129.0f);
    }
    __syncthreads();
    if (tid < 32) {
        const int o_dw0 = (int)P.sect[0][S_DW0];
        const f32x4 f0 = ldt(wrs, 2 * 34 * 6, o_dw0), f1 = ldt(wrs, 2 * 34 * 6 + 1, o_dw0);
        const float filt[7] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z};
        float mp[14];
#pragma unroll
        for (int t = 0; t < 8; ++t) mp[3 + t] = colmean[t * 32 + tid];
        mp[0] = mp[3 + 3]; mp[1] = mp[3 + 2]; mp[2] = mp[3 + 1];          // reverse(mean[1:4])
        mp[11] = mp[3 + 6]; mp[12] = mp[3 + 5]; mp[13] = mp[3 + 4];       // reverse(mean[-4:-1])
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 7; ++k) s = fmaf(filt[k], mp[t + k], s);
            acc += s;
        }
        mmv[tid] = acc * 0.125f;
    }
    __syncthreads();

    // ---- P2: first layer; wave w produces output column t' = w (input column t = 2w), 16 channels ---------
    {
        const int tcol = 2 * w;
        int o_dw0 = (int)P.sect[w][S_DW0], o_l0 = (int)P.sect[w][S_L0];
        asm volatile("" : "+s"(o_dw0), "+s"(o_l0));
        f32x16 acc = acc_of(WL(o_l0), WL(o_l0 + 1), WL(o_l0 + 2), WL(o_l0 + 3));
        const int ws = o_l0 + 4;
        const float mm = mmv[m];
        f32x4 Wa = WL(ws), Wb = WL(ws + 1), Wc = WL(ws + 2), Wd = WL(ws + 3);
#pragma unroll 1
        for (int j = 0; j < 17; ++j) {
            const int q = 2 * j + h;                 // this lane's channel quad (33 -> all zero)
            // depthwise k5 p2 over the 8 columns, magnitude part and normalised part, + the undelayed x1 quads
            f32x4 dm = ldt(wrs, (q * 6 + 5), o_dw0), dn = ldt(wrs, ((34 + q) * 6 + 5), o_dw0);   // biases
            f32x4 xm = zero4, xn = zero4;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = tcol + k - 2;
                if (tc >= 0 && tc < 8 && q < 33) {
                    const f32x4 mg = RX[(MAG_Q * tc + q) * QS + m];
                    const f32x4 sp = f32x4{log1p20(mg.x) - mm, log1p20(mg.y) - mm, log1p20(mg.z) - mm, log1p20(mg.w) - mm};
                    dm = fma4(ldt(wrs, q * 6 + k, o_dw0), mg, dm);
                    dn = fma4(ldt(wrs, (34 + q) * 6 + k, o_dw0), sp, dn);
                    if (k == 2) { xm = mg; xn = sp; }
                }
            }
            dm = relu4(dm);
            dn = relu4(dn);
            const int jn = j < 16 ? j + 1 : 16;
            const f32x4 nWa = WL(ws + 4 * jn), nWb = WL(ws + 4 * jn + 1), nWc = WL(ws + 4 * jn + 2), nWd = WL(ws + 4 * jn + 3);
            SB();
            TG_MMA(acc, Wa, Wb, Wc, Wd, dm, xm, dn, xn)      // pw|mag . dm + proj|mag . xm + pw|norm . dn + proj|norm . xn
            SB();
            Wa = nWa; Wb = nWb; Wc = nWc; Wd = nWd;
        }
        // rows 0..15 of the tile are the 16 channels: registers of g = 0,1
#pragma unroll
        for (int g = 0; g < 2; ++g) RX[(R_A16 + 4 * w + 2 * g) * QS + hq] = relu4(quad_of(acc, g));
    }
    __syncthreads();   // the magnitude rows are free from here on

    // previous h of both LSTM layers -> rows R_H0.. (32 quads per stream), requested now, used much later
    {
        const int fm = tid & 31, part = tid >> 5;
        const int g2 = tile0 + fm;
        const int s2 = g2 < P.n ? (P.slots ? P.slots[g2] : g2) : -1;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = part * 4 + qq;
            f32x4 v = zero4;
            if (s2 >= 0) v = reinterpret_cast<const f32x4 *>(P.state + (size_t)s2 * 256)[q];
            RX[(R_H0 + q) * QS + fm] = v;
        }
    }

    // ---- P3: s0 1x1 16 -> 16 on the 4 kept columns (stride 2 already applied: t = 0,2,4,6) -------------
    {
        int o = (int)P.sect[w][S_S0];
        asm volatile("" : "+s"(o));
        f32x16 acc = acc_of(WL(o), WL(o + 1), WL(o + 2), WL(o + 3));
        acc = mfma4(WL(o + 4), RX[(R_A16 + 4 * w + 0) * QS + hq], acc);
        acc = mfma4(WL(o + 5), RX[(R_A16 + 4 * w + 2) * QS + hq], acc);
#pragma unroll
        for (int g = 0; g < 2; ++g) RX[(R_Y0 + 4 * w + 2 * g) * QS + hq] = relu4(quad_of(acc, g));
    }
    __syncthreads();

    // ---- P4: block 1 (16 -> 32): dw k5 over the 4 columns (VALU, in registers) -> pw, + proj(y) ---------
    {
        int o = (int)P.sect[w][S_L1];
        asm volatile("" : "+s"(o));
        f32x16 acc = acc_of(WL(o + 1), WL(o + 2), WL(o + 3), WL(o + 4));
        f32x4 d[2], y[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 2 * j + h;
            d[j] = ldt(wrs, q * 6 + 5, o);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = w + k - 2;
                if (tc >= 0 && tc < 4) d[j] = fma4(ldt(wrs, q * 6 + k, o), RX[(R_Y0 + 4 * tc + 2 * j) * QS + hq], d[j]);
            }
            d[j] = relu4(d[j]);
            y[j] = RX[(R_Y0 + 4 * w + 2 * j) * QS + hq];
        }
        TG_MMA(acc, WL(o + 5), WL(o + 6), WL(o + 7), WL(o + 8), d[0], d[1], y[0], y[1])
        store_tile_relu(RX, R_Y1 + 8 * w, m, h, acc);
    }
    __syncthreads();

    // ---- P5: s1 1x1 32 -> 32, stride 2: columns 0 and 2; waves 0,1 -------------------------------------
    if (w < 2) {
        int o = (int)P.sect[w][S_S1];
        asm volatile("" : "+s"(o));
        f32x16 acc = acc_of(WL(o), WL(o + 1), WL(o + 2), WL(o + 3));
        const int r = R_Y1 + 8 * (2 * w);
        TG_MMA(acc, WL(o + 4), WL(o + 5), WL(o + 6), WL(o + 7), RX[(r + 0) * QS + hq], RX[(r + 2) * QS + hq], RX[(r + 4) * QS + hq], RX[(r + 6) * QS + hq])
        store_tile_relu(RX, R_Y2 + 8 * w, m, h, acc);
    }
    __syncthreads();

    // ---- P6: block 2 (32 -> 32, identity residual) on 2 columns; waves 0,1 ------------------------------
    if (w < 2) {
        int o = (int)P.sect[w][S_L2];
        asm volatile("" : "+s"(o));
        f32x16 acc = acc_of(WL(o + 1), WL(o + 2), WL(o + 3), WL(o + 4));
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 2 * j + h;
            d[j] = ldt(wrs, q * 6 + 5, o);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = w + k - 2;
                if (tc >= 0 && tc < 2) d[j] = fma4(ldt(wrs, q * 6 + k, o), RX[(R_Y2 + 8 * tc + 2 * j) * QS + hq], d[j]);
            }
            d[j] = relu4(d[j]);
        }
        TG_MMA(acc, WL(o + 5), WL(o + 6), WL(o + 7), WL(o + 8), d[0], d[1], d[2], d[3])
        // + identity residual: lane (m,h) register 4g+i is channel 8g+4h+i = quad 2g+h of the input
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 r = RX[(R_Y2 + 8 * w + 2 * g) * QS + hq], a = quad_of(acc, g);
            RX[(R_Y3 + 8 * w + 2 * g) * QS + hq] = relu4(f32x4{a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w});
        }
    }
    __syncthreads();

    // ---- P7: s2 1x1 32 -> 32, stride 2: column 0; wave 0 -------------------------------------------------
    if (w == 0) {
        int o = (int)P.sect[w][S_S2];
        asm volatile("" : "+s"(o));
        f32x16 acc = acc_of(WL(o), WL(o + 1), WL(o + 2), WL(o + 3));
        TG_MMA(acc, WL(o + 4), WL(o + 5), WL(o + 6), WL(o + 7), RX[(R_Y3 + 0) * QS + hq], RX[(R_Y3 + 2) * QS + hq], RX[(R_Y3 + 4) * QS + hq], RX[(R_Y3 + 6) * QS + hq])
        store_tile_relu(RX, R_Y4, m, h, acc);
    }
    __syncthreads();

    // ---- P8: block 3 (32 -> 64) on the single column (dw: centre tap only); waves 0,1 = output tile -----
    if (w < 2) {
        int o = (int)P.sect[w][S_L3];
        asm volatile("" : "+s"(o));
        const int ob = o + 1 + 12 * w;
        f32x16 acc = acc_of(WL(ob), WL(ob + 1), WL(ob + 2), WL(ob + 3));
        f32x4 d[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 2 * j + h;
            y[j] = RX[(R_Y4 + 2 * j) * QS + hq];
            d[j] = relu4(fma4(ldt(wrs, q * 6 + 2, o), y[j], ldt(wrs, q * 6 + 5, o)));
        }
        TG_MMA(acc, WL(ob + 4), WL(ob + 5), WL(ob + 6), WL(ob + 7), d[0], d[1], d[2], d[3])
        TG_MMA(acc, WL(ob + 8), WL(ob + 9), WL(ob + 10), WL(ob + 11), y[0], y[1], y[2], y[3])
        store_tile_relu(RX, R_Y5 + 8 * w, m, h, acc);
    }
    __syncthreads();

    // ---- P9: s3 1x1 64 -> 64; waves 0,1 = output tile ---------------------------------------------------
    if (w < 2) {
        int o = (int)P.sect[w][S_S3];
        asm volatile("" : "+s"(o));
        const int ob = o + 12 * w;
        f32x16 acc = acc_of(WL(ob), WL(ob + 1), WL(ob + 2), WL(ob + 3));
        TG_MMA(acc, WL(ob + 4), WL(ob + 5), WL(ob + 6), WL(ob + 7), RX[(R_Y5 + 0) * QS + hq], RX[(R_Y5 + 2) * QS + hq], RX[(R_Y5 + 4) * QS + hq], RX[(R_Y5 + 6) * QS + hq])
        TG_MMA(acc, WL(ob + 8), WL(ob + 9), WL(ob + 10), WL(ob + 11), RX[(R_Y5 + 8) * QS + hq], RX[(R_Y5 + 10) * QS + hq], RX[(R_Y5 + 12) * QS + hq], RX[(R_Y5 + 14) * QS + hq])
        store_tile_relu(RX, R_Y6 + 8 * w, m, h, acc);
    }
    __syncthreads();

    // ---- P10/P11: two stacked LSTM(64) cells.  wave w: unit half u = w&1; waves 0,1 contract the layer
    //      input (+bias), waves 2,3 contract h_{t-1}; partial gates meet in LDS, waves 0,1 finish the cell
    const int u = w & 1, kh = w >> 1;
    float part = 0.f;
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        int o = (int)P.sect[w][layer == 0 ? S_LSTM0 : S_LSTM1];
        asm volatile("" : "+s"(o));
        const int ob = o + 80 * u;
        f32x16 g4[4];
        int ws;
        const f32x4 *src;
        if (kh == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g4[q] = acc_of(WL(ob + 4 * q), WL(ob + 4 * q + 1), WL(ob + 4 * q + 2), WL(ob + 4 * q + 3));
            ws = ob + 16;
            src = RX + (layer == 0 ? R_Y6 : R_H0N) * QS + hq;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) g4[q] = (f32x16)(0.f);
            ws = ob + 48;
            src = RX + (layer == 0 ? R_H0 : R_H1) * QS + hq;
        }
        f32x4 Aw0 = WL(ws), Aw1 = WL(ws + 1), Aw2 = WL(ws + 2), Aw3 = WL(ws + 3), Aa = src[0], Bw0, Bw1, Bw2, Bw3, Ba;
#define LS_LD(S, it) S##w0 = WL(ws + 4 * (it)); S##w1 = WL(ws + 4 * (it) + 1); S##w2 = WL(ws + 4 * (it) + 2); S##w3 = WL(ws + 4 * (it) + 3); S##a = src[(2 * (it)) * QS];
#define LS_MMA(S) g4[0] = mfma4(S##w0, S##a, g4[0]); g4[1] = mfma4(S##w1, S##a, g4[1]); g4[2] = mfma4(S##w2, S##a, g4[2]); g4[3] = mfma4(S##w3, S##a, g4[3]);
        for (int it = 0; it < 8; it += 2) {
            LS_LD(B, it + 1) SB();
            LS_MMA(A) SB();
            const int itn = it + 2 < 8 ? it + 2 : 6;
            LS_LD(A, itn) SB();
            LS_MMA(B) SB();
        }
#undef LS_LD
#undef LS_MMA
        if (kh == 1) {
            float *gp = gpart + (size_t)u * 64 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) gp[(q * 16 + r) * 64] = g4[q][r];
        }
        __syncthreads();
        if (kh == 0) {
            const float *gp = gpart + (size_t)u * 64 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) g4[q][r] += gp[(q * 16 + r) * 64];
            int oh = (int)P.sect[w][S_HEADB];
            asm volatile("" : "+s"(oh));
            float *st = P.state + (size_t)slot * 256;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 i4 = quad_of(g4[0], g), f4 = quad_of(g4[1], g), c4g = quad_of(g4[2], g), o4 = quad_of(g4[3], g);
                const int unit = 32 * u + 8 * g + 4 * h;
                f32x4 cp = zero4;
                if (live) cp = *reinterpret_cast<const f32x4 *>(st + 128 + 64 * layer + unit);
                f32x4 cn, hn;
#define CELL(k)                                                             \
    cn.k = sigmoidf_(f4.k) * cp.k + sigmoidf_(i4.k) * tanhf_(c4g.k);      \
    hn.k = sigmoidf_(o4.k) * tanhf_(cn.k);
                CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
                if (live) {
                    *reinterpret_cast<f32x4 *>(st + 128 + 64 * layer + unit) = cn;
                    *reinterpret_cast<f32x4 *>(st + 64 * layer + unit) = hn;
                }
                if (layer == 0) {
                    RX[(R_H0N + 8 * u + 2 * g) * QS + hq] = hn;
                } else {
                    const f32x4 hw = WL(oh + 1 + 4 * u + g);
                    part += hw.x * fmaxf(hn.x, 0.f) + hw.y * fmaxf(hn.y, 0.f) + hw.z * fmaxf(hn.z, 0.f) + hw.w * fmaxf(hn.w, 0.f);
                }
            }
        }
        __syncthreads();
    }
    if (kh == 0) {
        part += __shfl_xor(part, 32);
        if (h == 0) headp[u * 32 + m] = part;
    }
    __syncthreads();

    // ---- head + state machine ------------------------------------------------------------------------
    if (tid < MT && tile0 + tid < P.n) {
        const float hb = P.wstream[(size_t)P.sect[0][S_HEADB] * BLK_FLOATS];
        const float p = fminf(sigmoidf_(hb + headp[tid] + headp[32 + tid]), 1.0f);
        const int sm_slot = P.slots ? P.slots[tile0 + tid] : tile0 + tid;
        P.probs[(size_t)(tile0 + tid) * T + tframe] = p;
        SmSlot sm = P.sm[sm_slot];
        int seg = 0;
        const int ev = sm_step(sm, p, &seg);
        P.sm[sm_slot] = sm;
        if (P.events) P.events[(size_t)(tile0 + tid) * T + tframe] = (uint8_t)ev;
        if (P.seg_frames) {
            if (ev & 2) P.seg_frames[tile0 + tid] = seg;
            else if (tframe == 0) P.seg_frames[tile0 + tid] = 0;
        }
    }
}

// host-callable launcher: the T frames of a call run as 2 T launches on one stream (state lives in HBM between them)
extern "C" hipError_t vadk_launch_silero_v4(const vadk::StepParams *p, hipStream_t stream) {
    const int tiles = (p->n + vadk::MT - 1) / vadk::MT;
    if (tiles <= 0) return hipSuccess;
    for (int t = 0; t < p->T; ++t) {
        hipLaunchKernelGGL(silero_v4_stft, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p, t);
        hipLaunchKernelGGL(silero_v4_tail, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p, t);
    }
    return hipGetLastError();
}

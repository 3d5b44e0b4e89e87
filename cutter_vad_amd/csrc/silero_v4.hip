// Fused Silero-VAD V4 (16 kHz) step kernels for MI355X / gfx950.
//
// Replaces `session.run` on silero_vad.onnx's 16 kHz branch for a batch of independent streams
// (reference call site: /root/reference/src/real_time_vad/core/silero_model.py:433 with the feeds
// of :494-499; dataflow: SURVEY.md §8 a8), plus the per-frame pre-steps and the state machine, as
// in silero_v5.hip.  One launch per frame, two parts that time-share the CU's LDS (vad_layout.h has the two layouts):
//
//   STFT part : load + gate + reflect-pad(96,96) + window + 4-way fold -> 8-column DFT (MFMA, K = 64) -> |.|, kept in
//               REGISTERS (128 per lane: the 258 x 8 first-layer input of 32 streams does not fit LDS next to the STFT
//               operands, but one wave per SIMD owns 512 registers)
//   tail      : the magnitudes move into LDS once the STFT operands are dead, then log(1 + |X| 2^20), adaptive
//               normalisation, first layer (dw k5 + pw + proj), 1x1 stride convs, 3 separable blocks, LSTM(64) x 2,
//               head, state machine
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

// -DVADK_STAMPS (tools/kbench4.cpp): s_memtime at phase boundaries, [block][wave][32]; the STFT part uses 0..15, the tail 16..31
#ifdef VADK_STAMPS
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (lane == 0) P.stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + (k)] = clock64();          \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

// one "thin" MFMA group: 4 k-iterations = 4 weight blocks x 4 activation quads -> 16 MFMAs on one accumulator
#define TG_MMA(acc, w0, w1, w2, w3, a0, a1, a2, a3) \
    acc = mfma4(w0, a0, acc); acc = mfma4(w1, a1, acc); acc = mfma4(w2, a2, acc); acc = mfma4(w3, a3, acc);

__device__ __forceinline__ f32x4 ldt(__amdgpu_buffer_rsrc_t rs, int row, int blk) {   // table row (float4) of a VALU table
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, row * 16, blk * 1024, 0));
}

// 16 output rows x 16 streams x 16 channels: four v_mfma_f32_16x16x4_f32 (component i = channels 4 kq + i of the lane groups kq)
__device__ __forceinline__ f32x4 mfma16(f32x4 w, f32x4 a, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) {
    return f32x4{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}

// a double moved across lanes inside each group of 4 (DPP quad_perm on its two halves)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float log1p20(float mag) {   // log(1 + mag * 2^20): Mul, Add, Log of the graph
    return __builtin_amdgcn_logf(1.0f + mag * 1048576.0f) * 0.69314718055994531f;
}
// the same minus the adaptive-normalisation mean: Sub of the graph
__device__ __forceinline__ float lognorm(float mag, float mm) { return log1p20(mag) - mm; }

}  // namespace

// K8: the graph's 8 kHz sub-model (else-branch, taken for every sr != 16000): identical up to block 2, then
// the third stride conv has stride 1 (Conv_632), so TWO columns go through block 3 and the last 1x1 conv, the LSTMs run
// two sequential time steps and the probability is the mean of the two sigmoids (ReduceMean over T).
template <bool K8>
__global__ void __launch_bounds__(NTHREADS, 1) silero_v4_step(const StepParams P, const int tframe) {
    using namespace vadk::v4;
    __shared__ f32x4 lds[V4_LDS_F4];
    // =================================================================================================
    //  STFT part
    // =================================================================================================
    // xp = reflect-pad(x, 96, 96) (numpy 'reflect'), materialised once per frame: [32 streams][176 quads].  x sits at
    // quads 24..151 (aligned: 96 = 4 * 24); the two 24-quad edges are mirrored from it after the load.  The fold then
    // reads plain quads (a branchy on-the-fly reflection serialised ~100 dependent LDS reads per group).
    f32x4 *const XP = lds;
    f32x4 *const UV = lds + K1_XS_F4;
    constexpr int XPQ = K1_XP_QUADS;               // u/v of the two columns in flight: rows 64c' + q | 64c' + 32 + q
    f32x4 *const WT = UV + K1_UV_ROWS * QS;         // w[n], 64 quads (the stored basis' k = 0 row)
    float *const nyqv = reinterpret_cast<float *>(WT + K1_WT_F4);   // [2][32] |X128| of the two columns in flight
    float *const dcv = nyqv + 64;                   // [2][32] X0 (signed)
    float *const fcor = dcv + 64;                   // [2 columns][y128, a64, b64][32 streams]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int hq = h * QS + m;
    const int tile0 = blockIdx.x * MT;
    const int T = P.T;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const int lane16 = lane * 16;
    const int o_stft = (int)P.sect[w][S_STFT];

    // the tail's view of the same LDS (second layout) and its per-lane constants
    f32x4 *const RX = lds;
    float *const misc = reinterpret_cast<float *>(lds + K2_ROWS * QS);
    float *const mmv = misc;                 // [32]  mean_mean per stream
    float *const colmean = misc + 32;        // [8][32]
    float *const headp = misc + 32 + 256;    // [2][32]
    float *const colpart = misc + K2_MISC_FLOATS;   // [4 waves][8 columns][32 streams]: partial log sums
    const int gf = tile0 + m;
    const bool live = gf < P.n;
    const int slot = live ? (P.slots ? P.slots[gf] : gf) : 0;
#define WL(blk) ldw(wrs, lane16, (blk))
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    STAMP(0);
    // ---- raw frame -> LDS (gate + int16 scaling fused), lanes run along the frame ---------------
    // All 16 quads of a thread are requested before the first is used (one HBM round trip, not one per batch); the
    // buffer descriptor's range check zero-fills the streams past n.
    {
        const float thr = P.thresh;
        const bool f32in = P.fmt == 0;
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void *>(P.frames), 0, (int)((unsigned)P.n * (unsigned)T * (f32in ? 2048u : 1024u)), 0x00020000);
        if (f32in) {
            u32x4 xv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int idx = it * NTHREADS + tid;
                xv[it] = __builtin_amdgcn_raw_buffer_load_b128(frs, (((tile0 + (idx >> 7)) * T + tframe) * 128 + (idx & 127)) * 16, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int idx = it * NTHREADS + tid;
                XP[(idx >> 7) * XPQ + 24 + (idx & 127)] = gate4(__builtin_bit_cast(f32x4, xv[it]), thr);
            }
        } else {
            const float sc = P.fmt == 1 ? 32767.0f : 32768.0f, rsc = 1.0f / sc;
            u32x2 sv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int idx = it * NTHREADS + tid;
                sv[it] = __builtin_amdgcn_raw_buffer_load_b64(frs, (((tile0 + (idx >> 7)) * T + tframe) * 128 + (idx & 127)) * 8, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int idx = it * NTHREADS + tid;
                const int s0 = (int)(short)(sv[it].x & 0xffffu), s1 = (int)(short)(sv[it].x >> 16);
                const int s2 = (int)(short)(sv[it].y & 0xffffu), s3 = (int)(short)(sv[it].y >> 16);
                XP[(idx >> 7) * XPQ + 24 + (idx & 127)] = gate4(f32x4{i16_div(s0, sc, rsc), i16_div(s1, sc, rsc), i16_div(s2, sc, rsc), i16_div(s3, sc, rsc)}, thr);
            }
        }
    }
    if (tid < K1_WT_F4) WT[tid] = ldw(wrs, tid * 16, (int)P.sect[0][S_NYQ]);
    __syncthreads();
    // mirrored edges: xp[i] = x[96 - i] (i < 96) and xp[608 + i] = x[510 - i]; in xp coordinates x[k] = xp[96 + k].
    //   left  quad Q  < 24 : {x[96-4Q], x[95-4Q], x[94-4Q], x[93-4Q]} = {lo.x, hi.w, hi.z, hi.y}, lo = xp quad 48-Q, hi = 47-Q
    //   right quad Q >= 152: x[1118-4Q-j], j = 0..3                    = {p.z, p.y, p.x, pm.w},  p  = xp quad 303-Q, pm = 302-Q
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = it * NTHREADS + tid;          // 32 streams x 48 edge quads
        const int ms = idx / 48, k = idx - ms * 48;
        f32x4 *row = XP + ms * XPQ;
        if (k < 24) {
            const f32x4 lo = row[48 - k], hi = row[47 - k];
            row[k] = f32x4{lo.x, hi.w, hi.z, hi.y};
        } else {
            const int Q = 128 + k;
            const f32x4 p = row[303 - Q], pm = row[302 - Q];
            row[Q] = f32x4{p.z, p.y, p.x, pm.w};
        }
    }
    __syncthreads();
    STAMP(1);

    // ---- early requests: everything that comes from HBM and is needed late is asked for now,
    //      while the frame is being folded: the slot's state machine
    //      (-> LDS, past both layouts), c_{t-1} of both LSTM layers (registers), head weights
    SmSlot *const smL = reinterpret_cast<SmSlot *>(lds + V4_SM_F4);
    // (tid & 31 == m: ONE slot lookup serves h, c and the state machine.  The 96 B are requested here by every thread for
    // the stream tid & 31 and stored to LDS by threads < 32 at the start of the tail: stored right here, the wait for them
    // was a memory round trip in front of the first fold.)
    const bool sm_thread = tid < MT && live;
    const int sm_slot = slot;
    f32x4 smq[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) smq[k] = reinterpret_cast<const f32x4 *>(P.sm + slot)[k];
    const int u = w & 1;                           // output tile of the 64-channel phases P8 / P9
    f32x4 cprev[2][2], hwq[2];                     // LSTM: wave w owns units 16w .. 16w+15; a lane holds 8 of them (2 quads)
    {
        const float *st = P.state + (size_t)slot * 256;
        int oh = (int)P.sect[w][S_HEADB];
        asm volatile("" : "+s"(oh));
#pragma unroll
        for (int layer = 0; layer < 2; ++layer)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(st + 128 + 64 * layer + 16 * w + 8 * e + 4 * h);
                cprev[layer][e] = live ? v : zero4;
            }
#pragma unroll
        for (int e = 0; e < 2; ++e) hwq[e] = WL(oh + 1 + 4 * (w >> 1) + 2 * (w & 1) + e);
    }
    f32x4 hprev[4];                                // h_{t-1} of both layers, stream tid & 31, quads 4 (tid >> 5) .. + 3
    {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(P.state + (size_t)slot * 256)[(tid >> 5) * 4 + qq];
            hprev[qq] = live ? v : zero4;
        }
    }
    const float hb = P.wstream[(size_t)P.sect[0][S_HEADB] * BLK_FLOATS];


    // window of the stored basis (its k = 0 cosine row): w[n] and w[128 + n] = w[128 - n] for this lane's n = 4q .. 4q + 3
    const int fq = tid & 15;
    const int o_win = (int)P.sect[w][S_NYQ];
    const f32x4 W1 = ldw(wrs, fq * 16, o_win), W3 = ldw(wrs, (32 + fq) * 16, o_win);
    const float w64 = ldw(wrs, 16 * 16, o_win).x;                 // w[64] = w[192]

    f32x4 mg[8][4];                                  // |X| of this wave's 32 bins, 8 columns: register 4g+i = bin row 8g+4h+i
    float nyq[4];                                    // threads < 64: |X[128]| of column 2 grp + h, stream m
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {              // STFT columns 2 grp, 2 grp + 1 (hop 64 on the padded frame)
        int ws = o_stft;
        asm volatile("" : "+s"(ws));
        // ---- window + 4-way fold (as silero_v5.hip): y = w . xp[64 t ..]; for n = 4q .. 4q + 3, q = 0..15:
        //      pe, po = (y[n] + y[256-n]) +- (y[128-n] + y[128+n]);  qe, qo = (y[n] - y[256-n]) -+ (y[128-n] - y[128+n])
        //      rows 64 c' + {0, 16, 32, 48} + q;  n = 0, 64, 128 enter as rank-1 corrections after the MFMAs
        {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o = rr * 16 + (tid >> 4);      // (column c', stream): o = c' * 32 + stream
                const int cp = o >> 5, ms = o & 31;
                const int Q0 = 16 * (2 * grp + cp);      // first xp quad of the column
                const f32x4 *xs = XP + ms * XPQ + Q0;
                const f32x4 xA = xs[fq], xC = xs[32 + fq];
                const f32x4 r1a = xs[32 - fq], r1b = xs[31 - fq];
                const f32x4 r2a = xs[fq == 0 ? 63 : 64 - fq], r2b = xs[63 - fq];   // quad 64 of the last column does not exist; its only use (n = 0) is masked
                const f32x4 y1 = f32x4{xA.x * W1.x, xA.y * W1.y, xA.z * W1.z, xA.w * W1.w};
                const f32x4 y3 = f32x4{xC.x * W3.x, xC.y * W3.y, xC.z * W3.z, xC.w * W3.w};
                const f32x4 y2 = f32x4{r1a.x * W3.x, r1b.w * W3.y, r1b.z * W3.z, r1b.y * W3.w};
                const f32x4 y4 = f32x4{r2a.x * W1.x, r2b.w * W1.y, r2b.z * W1.z, r2b.y * W1.w};
                const f32x4 s14 = f32x4{y1.x + y4.x, y1.y + y4.y, y1.z + y4.z, y1.w + y4.w};
                const f32x4 d14 = f32x4{y1.x - y4.x, y1.y - y4.y, y1.z - y4.z, y1.w - y4.w};
                const f32x4 s23 = f32x4{y2.x + y3.x, y2.y + y3.y, y2.z + y3.z, y2.w + y3.w};
                const f32x4 d23 = f32x4{y2.x - y3.x, y2.y - y3.y, y2.z - y3.z, y2.w - y3.w};
                f32x4 pe = f32x4{s14.x + s23.x, s14.y + s23.y, s14.z + s23.z, s14.w + s23.w};
                f32x4 po = f32x4{s14.x - s23.x, s14.y - s23.y, s14.z - s23.z, s14.w - s23.w};
                f32x4 qe = f32x4{d14.x - d23.x, d14.y - d23.y, d14.z - d23.z, d14.w - d23.w};
                f32x4 qo = f32x4{d14.x + d23.x, d14.y + d23.y, d14.z + d23.z, d14.w + d23.w};
                if (fq == 0) {
                    pe.x = po.x = qe.x = qo.x = 0.f;
                    const float y64 = xs[16].x * w64, y192 = xs[48].x * w64;
                    fcor[(cp * 3 + 0) * 32 + ms] = y3.x;           // y[128]
                    fcor[(cp * 3 + 1) * 32 + ms] = y64 + y192;     // a64
                    fcor[(cp * 3 + 2) * 32 + ms] = y64 - y192;     // b64
                }
                st2(&UV[(64 * cp + fq) * QS + ms], pe);
                st2(&UV[(64 * cp + 16 + fq) * QS + ms], po);
                st2(&UV[(64 * cp + 32 + fq) * QS + ms], qe);
                st2(&UV[(64 * cp + 48 + fq) * QS + ms], qo);
            }
        }
        // ---- the two REAL bins, k = 0 and k = 128, in float64 straight from the samples: X0 = E + O, X128 = E - O with
        //      E / O = sum over even / odd n of w[n] xp[n] (the products of two floats are exact in double, the sums carry
        //      ~1e-16).  These are the inputs on which log(1 + |X| 2^20) is ill-conditioned: a real sum of 256 terms of size
        //      ~0.1 lands within 1e-5 of zero about once in 10^4 columns, and there an fp32 accumulation error of 4e-7 moves
        //      the log by 0.1 and the probability by 6e-4 (tools/v4_real_bins.py; a complex bin needs re and im to cancel
        //      together).  64 (column, stream) pairs, 4 lanes each; lane `part` sums samples 64 part .. 64 part + 63,
        //      quad order XOR-swizzled by (stream & 3, part) so that 16 neighbouring lanes read 16 different banks.
        {
            const int pair = tid >> 2, part = tid & 3;
            const int cp = pair >> 5, ms = pair & 31;
            const unsigned sw = (unsigned)(4 * (ms & 3) + part);
            const unsigned bx = ((unsigned)(ms * XPQ + 16 * (2 * grp + cp) + 16 * part) ^ sw) << 4;   // byte offsets; both bases are multiples of 16 quads
            const unsigned bw = ((unsigned)(K1_XS_F4 + K1_UV_ROWS * QS + 16 * part) ^ sw) << 4;
            const char *const lb = reinterpret_cast<const char *>(lds);
            double e0 = 0., e1 = 0., o0 = 0., o1 = 0.;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(lb + (bx ^ (unsigned)(i << 4)));
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(lb + (bw ^ (unsigned)(i << 4)));
                e0 = __builtin_fma((double)wv.x, (double)xv.x, e0);
                o0 = __builtin_fma((double)wv.y, (double)xv.y, o0);
                e1 = __builtin_fma((double)wv.z, (double)xv.z, e1);
                o1 = __builtin_fma((double)wv.w, (double)xv.w, o1);
            }
            double e = e0 + e1, o = o0 + o1;
            e += dpp_f64<0xB1>(e); o += dpp_f64<0xB1>(o);      // quad_perm [1,0,3,2]
            e += dpp_f64<0x4E>(e); o += dpp_f64<0x4E>(o);      // quad_perm [2,3,0,1]
            if (part == 0) {
                dcv[cp * 32 + ms] = (float)(e + o);
                nyqv[cp * 32 + ms] = fabsf((float)(e - o));
            }
        }
        f32x4 Are = ldw(wrs, lane16, ws), Aim = ldw(wrs, lane16, ws + 1);
        SB();
        if (grp == 0) STAMP(2);
        __syncthreads();
        if (grp == 0) STAMP(3);
        // ---- MFMA: wave w = bins bin_of_channel(32 w + r): cos on pe | po, -sin on qe | qo (even | odd bins), two columns, K = 64
        // the accumulators start from the rank-1 terms of n = 0, 64, 128 (register 4g+i holds tile row r = 8g+4h+i: (-1)^r = (-1)^i)
        //   even bins: re += y128 + a64 (-1)^r ; odd bins: re -= y128, im -= b64 (-1)^r
        f32x16 are[2], aim[2];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const float y128 = fcor[(cp * 3 + 0) * 32 + m], a64 = fcor[(cp * 3 + 1) * 32 + m], b64 = fcor[(cp * 3 + 2) * 32 + m];
            const float rp = w < 2 ? y128 + a64 : -y128, rm = w < 2 ? y128 - a64 : -y128;
            const float ip = w < 2 ? 0.f : -b64, im_ = w < 2 ? 0.f : b64;
            const f32x4 qr = f32x4{rp, rm, rp, rm}, qi = f32x4{ip, im_, ip, im_};
            are[cp] = acc_of(qr, qr, qr, qr);
            aim[cp] = acc_of(qi, qi, qi, qi);
        }
        {
            const int rR = w < 2 ? 0 : 16, rI = w < 2 ? 32 : 48;
            const f32x4 *const XR = UV + rR * QS + hq, *const XI = UV + rI * QS + hq;
            f32x4 Au0 = XR[0], Au1 = XR[64 * QS], Av0 = XI[0], Av1 = XI[64 * QS];
            f32x4 Bre, Bim, Bu0, Bu1, Bv0, Bv1;
#define K1_LD(S, jj)                                                                        \
    S##re = ldw(wrs, lane16, ws + 2 * (jj)); S##im = ldw(wrs, lane16, ws + 2 * (jj) + 1);   \
    S##u0 = XR[(2 * (jj)) * QS]; S##u1 = XR[(64 + 2 * (jj)) * QS];                          \
    S##v0 = XI[(2 * (jj)) * QS]; S##v1 = XI[(64 + 2 * (jj)) * QS];
#define K1_MMA(S)                                                                           \
    are[0] = mfma4(S##re, S##u0, are[0]); are[1] = mfma4(S##re, S##u1, are[1]);             \
    aim[0] = mfma4(S##im, S##v0, aim[0]); aim[1] = mfma4(S##im, S##v1, aim[1]);
            for (int j = 0; j < 8; j += 2) {
                K1_LD(B, j + 1) SB();
                K1_MMA(A) SB();
                const int jn = j + 2 < 8 ? j + 2 : 6;
                K1_LD(A, jn) SB();
                K1_MMA(B) SB();
            }
#undef K1_LD
#undef K1_MMA
        }
        if (grp == 0) STAMP(4);
        {   // bin 0 (wave 0, tile row 0 = register 0 of the lower half-wave) takes the float64 sum; its im is identically 0
            const bool own0 = (w == 0) & (h == 0);
            const float d0 = dcv[m], d1 = dcv[32 + m];
            are[0].s0 = own0 ? d0 : are[0].s0;
            are[1].s0 = own0 ? d1 : are[1].s0;
        }
        // ---- magnitudes: kept in registers until the STFT operands in LDS are dead
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 r = quad_of(are[cp], g), i = quad_of(aim[cp], g);
                mg[2 * grp + cp][g] = f32x4{mag_(r.x, i.x), mag_(r.y, i.y), mag_(r.z, i.z), mag_(r.w, i.w)};
            }
        }
        if (grp == 0) STAMP(5);
        nyq[grp] = nyqv[(tid < 64 ? h : 0) * 32 + m];
        __syncthreads();       // every wave done with UV / fcor / dcv before the next fold overwrites them
        if (grp == 0) STAMP(6);
        if (grp == 3) STAMP(7);
    }


    // =================================================================================================
    //  tail: the LDS is re-used with the second layout from here on (every wave is past the last group's barrier)
    // =================================================================================================
    STAMP(16);
    if (tid < MT) {
#pragma unroll
        for (int k = 0; k < 6; ++k) reinterpret_cast<f32x4 *>(smL + tid)[k] = smq[k];
    }
    // ---- P0 + P1: magnitudes registers -> LDS rows (33 t + q), and the per-column mean of the log-spectrum on the way:
    //      a lane sums the logs of its 16 bins per column, the two half-waves meet in a shuffle, the four waves in LDS
#pragma unroll
    for (int tc = 0; tc < 8; ++tc) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = mg[tc][g];
            RX[(MAG_Q * tc + 8 * w + 2 * g + h) * QS + m] = v;
            s += (log1p20(v.x) + log1p20(v.y)) + (log1p20(v.z) + log1p20(v.w));
        }
        s += __shfl_xor(s, 32);
        if (h == 0) colpart[(w * 8 + tc) * 32 + m] = s;
    }
    if (tid < 64) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) RX[(MAG_Q * (2 * grp + h) + 32) * QS + m] = f32x4{nyq[grp], 0.f, 0.f, 0.f};
    }
    __syncthreads();
    {
        const int ms = tid & 31, tc = tid >> 5;
        const float s = ((colpart[tc * 32 + ms] + colpart[(8 + tc) * 32 + ms]) + (colpart[(16 + tc) * 32 + ms] + colpart[(24 + tc) * 32 + ms])) +
                        log1p20(RX[(MAG_Q * tc + 32) * QS + ms].x);
        colmean[tc * 32 + ms] = s * (1.0f / 129.0f);
    }
    STAMP(17);
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

    // Weights and depthwise tables of the short phases P3..P9 and of the LSTM layers are requested one or two phases
    // before they are used (sets A, B, C, L): every phase is only 16..32 MFMAs long, an L2 round trip at its start
    // would cost more than the phase itself.  One wave per SIMD owns 512 registers, so holding them is free.
    f32x4 p3w[6], p4w[8], p4t[12], p5w[8];       // set A: requested before the P2 loop
    f32x4 p6w[8], p6t[24], p7w[8];               // set B: requested at P3
    f32x4 p8w[12], p8t[8], p9w[12];              // set C: requested at P5
    f32x4 lb[8], lw[4];                          // set L: LSTM gate biases (tiles A, B) + the first two k-iterations' weights, per layer
#define PRE_A                                                                                     \
    {                                                                                             \
        int o_ = (int)P.sect[w][S_S0];                                                            \
        _Pragma("unroll") for (int k = 0; k < 6; ++k) p3w[k] = WL(o_ + k);                        \
        o_ = (int)P.sect[w][S_L1];                                                                \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) p4w[k] = WL(o_ + 1 + k);                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
            _Pragma("unroll") for (int k = 0; k < 6; ++k) p4t[6 * j + k] = ldt(wrs, (2 * j + h) * 6 + k, o_); \
        o_ = (int)P.sect[w][S_S1];                                                                \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) p5w[k] = WL(o_ + k);                        \
    }
#define PRE_B                                                                                     \
    {                                                                                             \
        int o_ = (int)P.sect[w][S_L2];                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) p6w[k] = WL(o_ + 1 + k);                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
            _Pragma("unroll") for (int k = 0; k < 6; ++k) p6t[6 * j + k] = ldt(wrs, (2 * j + h) * 6 + k, o_); \
        o_ = (int)P.sect[w][S_S2];                                                                \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) p7w[k] = WL(o_ + k);                        \
    }
#define PRE_C                                                                                     \
    {                                                                                             \
        int o_ = (int)P.sect[w][S_L3];                                                            \
        _Pragma("unroll") for (int k = 0; k < 12; ++k) p8w[k] = WL(o_ + 1 + 12 * u + k);          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            p8t[2 * j] = ldt(wrs, (2 * j + h) * 6 + 2, o_);                                       \
            p8t[2 * j + 1] = ldt(wrs, (2 * j + h) * 6 + 5, o_);                                   \
        }                                                                                         \
        o_ = (int)P.sect[w][S_S3];                                                                \
        _Pragma("unroll") for (int k = 0; k < 12; ++k) p9w[k] = WL(o_ + 12 * u + k);              \
    }
    // LSTM layer `layer`, this wave's 16 units: biases of tiles A (i|f) and B (g|o), weights of k-iterations 0 and 1
#define PRE_L(layer)                                                                              \
    {                                                                                             \
        const int ob_ = (int)P.sect[w][(layer) == 0 ? S_LSTM0 : S_LSTM1];                         \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) lb[k] = WL(ob_ + k);                        \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) lw[k] = WL(ob_ + 8 + k);                    \
    }

    // ---- P2: first layer (258 -> 16 channels), all four kept output columns t' (input column t = 2 t') in every wave, K
    //      split over the waves: wave w contracts the channel quads 16 j + 4 kq of j = w and w + 4 (128 bins; the Nyquist
    //      channel is a rank-1 term on the VALU, output column w in wave w).  The log-spectrum of a (channel quad, column)
    //      is evaluated once and serves the five taps of every output column (a column split makes each wave re-evaluate
    //      its neighbours' columns).  16 output channels = v_mfma_f32_16x16x4_f32 tiles (a 32-row tile wastes half of every
    //      MFMA): lane (n, kq) = (stream n | n + 16 by half sg, channel quad 4 j + kq); D: lane (n, rq) = output quad rq.
    //      The four partial tiles per column meet in LDS once the magnitude rows are dead.
    {
        const int n16 = lane & 15, kq = lane >> 4;
        int o_dw0 = (int)P.sect[w][S_DW0], o_l0 = (int)P.sect[w][S_L0];
        asm volatile("" : "+s"(o_dw0), "+s"(o_l0));
        f32x4 acc[4][2];
        {
            const f32x4 b0 = WL(o_l0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = w == 0 ? b0 : zero4;       // the bias enters once
        }
        const int ws = o_l0 + 5;
        const float mm2[2] = {mmv[n16], mmv[16 + n16]};
        // depthwise taps + bias of this lane's channel quad: 12 table rows per k-iteration, both iterations requested now
        f32x4 tb[2][12], wq[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q_ = 4 * (w + 4 * it) + kq;
            tb[it][0] = ldt(wrs, q_ * 6 + 5, o_dw0); tb[it][1] = ldt(wrs, (34 + q_) * 6 + 5, o_dw0);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                tb[it][2 + 2 * k] = ldt(wrs, q_ * 6 + k, o_dw0);
                tb[it][3 + 2 * k] = ldt(wrs, (34 + q_) * 6 + k, o_dw0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) wq[it][k] = WL(ws + 4 * (w + 4 * it) + k);
        }
        // Nyquist channel: quad 32 of the tables (component 0), its four weight columns in the D layout
        f32x4 tn[12], wn[4];
        tn[0] = ldt(wrs, 32 * 6 + 5, o_dw0); tn[1] = ldt(wrs, (34 + 32) * 6 + 5, o_dw0);
#pragma unroll
        for (int k = 0; k < 5; ++k) { tn[2 + 2 * k] = ldt(wrs, 32 * 6 + k, o_dw0); tn[3 + 2 * k] = ldt(wrs, (34 + 32) * 6 + k, o_dw0); }
#pragma unroll
        for (int k = 0; k < 4; ++k) wn[k] = WL(o_l0 + 1 + k);
        // weights of the next phases (P3, P4, P5): they do not depend on LDS and land during this loop
        PRE_A
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                const int ms = n16 + 16 * sg;
                const float mm = mm2[sg];
                f32x4 mgc[8], spc[8];
#pragma unroll
                for (int tc = 0; tc < 8; ++tc) mgc[tc] = RX[(MAG_Q * tc + 4 * (w + 4 * it) + kq) * QS + ms];
                SB();
#pragma unroll
                for (int tc = 0; tc < 8; ++tc)
                    spc[tc] = f32x4{lognorm(mgc[tc].x, mm), lognorm(mgc[tc].y, mm), lognorm(mgc[tc].z, mm), lognorm(mgc[tc].w, mm)};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // depthwise k5 p2 around input column 2c, magnitude part and normalised part
                    f32x4 dm = tb[it][0], dn = tb[it][1];        // biases
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const int tc = 2 * c + k - 2;
                        if (tc >= 0 && tc < 8) {
                            dm = fma4(tb[it][2 + 2 * k], mgc[tc], dm);
                            dn = fma4(tb[it][3 + 2 * k], spc[tc], dn);
                        }
                    }
                    dm = relu4(dm);
                    dn = relu4(dn);
                    SB();
                    // pw|mag . dm + proj|mag . x1 + pw|norm . dn + proj|norm . x1 (x1 = the undelayed column 2c)
                    f32x4 a_ = acc[c][sg];
                    a_ = mfma16(wq[it][0], dm, a_);
                    a_ = mfma16(wq[it][1], mgc[2 * c], a_);
                    a_ = mfma16(wq[it][2], dn, a_);
                    a_ = mfma16(wq[it][3], spc[2 * c], a_);
                    acc[c][sg] = a_;
                    SB();
                }
            }
        }
        // Nyquist channel of output column w (input column 2w): scalars per stream, rank-1 into this lane's output quad
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const int ms = n16 + 16 * sg;
            const float mm = mm2[sg];
            float dm = tn[0].x, dn = tn[1].x, xm = 0.f, xn = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = 2 * w + k - 2;
                if (tc >= 0 && tc < 8) {             // wave-uniform
                    const float mg = RX[(MAG_Q * tc + 32) * QS + ms].x;
                    const float sp = lognorm(mg, mm);
                    dm = fmaf(tn[2 + 2 * k].x, mg, dm);
                    dn = fmaf(tn[3 + 2 * k].x, sp, dn);
                    if (k == 2) { xm = mg; xn = sp; }
                }
            }
            dm = fmaxf(dm, 0.f);
            dn = fmaxf(dn, 0.f);
            const f32x4 r1 = wn[0] * dm + wn[1] * xm + wn[2] * dn + wn[3] * xn;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c == w) acc[c][sg] = acc[c][sg] + r1;
        }
        STAMP(18);
        __syncthreads();   // every wave is done with the magnitude rows: they now carry the partial tiles
        // PART[wave][column][output quad][stream]: the D layout of a 16x16 tile is exactly an LDS quad per lane
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) RX[((w * 4 + c) * 4 + kq) * QS + n16 + 16 * sg] = acc[c][sg];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * NTHREADS + tid;      // (column, channel quad, stream)
            const int cq = idx >> 5, ms = idx & 31;
            const f32x4 p0 = RX[cq * QS + ms], p1 = RX[(16 + cq) * QS + ms], p2 = RX[(32 + cq) * QS + ms], p3 = RX[(48 + cq) * QS + ms];
            RX[(R_A16 + cq) * QS + ms] = relu4((p0 + p1) + (p2 + p3));
        }
    }
    STAMP(19);
    __syncthreads();   // the magnitude rows are free from here on
    STAMP(20);

    // previous h of both LSTM layers (requested at kernel start) -> rows R_H0.. (32 quads per stream)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) RX[(R_H0 + (tid >> 5) * 4 + qq) * QS + (tid & 31)] = hprev[qq];

    // ---- P3: s0 1x1 16 -> 16 on the 4 kept columns (stride 2 already applied: t = 0,2,4,6) -------------
    {
        PRE_B
        SB();
        f32x16 acc = acc_of(p3w[0], p3w[1], p3w[2], p3w[3]);
        acc = mfma4(p3w[4], RX[(R_A16 + 4 * w + 0) * QS + hq], acc);
        acc = mfma4(p3w[5], RX[(R_A16 + 4 * w + 2) * QS + hq], acc);
#pragma unroll
        for (int g = 0; g < 2; ++g) RX[(R_Y0 + 4 * w + 2 * g) * QS + hq] = relu4(quad_of(acc, g));
    }
    __syncthreads();

    // ---- P4: block 1 (16 -> 32): dw k5 over the 4 columns (VALU, in registers) -> pw, + proj(y) ---------
    {
        f32x16 acc = acc_of(p4w[0], p4w[1], p4w[2], p4w[3]);
        f32x4 d[2], y[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            d[j] = p4t[6 * j + 5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = w + k - 2;
                if (tc >= 0 && tc < 4) d[j] = fma4(p4t[6 * j + k], RX[(R_Y0 + 4 * tc + 2 * j) * QS + hq], d[j]);
            }
            d[j] = relu4(d[j]);
            y[j] = RX[(R_Y0 + 4 * w + 2 * j) * QS + hq];
        }
        TG_MMA(acc, p4w[4], p4w[5], p4w[6], p4w[7], d[0], d[1], y[0], y[1])
        store_tile_relu(RX, R_Y1 + 8 * w, m, h, acc);
    }
    __syncthreads();

    STAMP(21);
    // ---- P5: s1 1x1 32 -> 32, stride 2: columns 0 and 2; waves 0,1 -------------------------------------
    PRE_C
    SB();
    if (w < 2) {
        f32x16 acc = acc_of(p5w[0], p5w[1], p5w[2], p5w[3]);
        const int r = R_Y1 + 8 * (2 * w);
        TG_MMA(acc, p5w[4], p5w[5], p5w[6], p5w[7], RX[(r + 0) * QS + hq], RX[(r + 2) * QS + hq], RX[(r + 4) * QS + hq], RX[(r + 6) * QS + hq])
        store_tile_relu(RX, R_Y2 + 8 * w, m, h, acc);
    }
    __syncthreads();

    // ---- P6: block 2 (32 -> 32, identity residual) on 2 columns; waves 0,1 ------------------------------
    if (w < 2) {
        f32x16 acc = acc_of(p6w[0], p6w[1], p6w[2], p6w[3]);
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[j] = p6t[6 * j + 5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = w + k - 2;
                if (tc >= 0 && tc < 2) d[j] = fma4(p6t[6 * j + k], RX[(R_Y2 + 8 * tc + 2 * j) * QS + hq], d[j]);
            }
            d[j] = relu4(d[j]);
        }
        TG_MMA(acc, p6w[4], p6w[5], p6w[6], p6w[7], d[0], d[1], d[2], d[3])
        // + identity residual: lane (m,h) register 4g+i is channel 8g+4h+i = quad 2g+h of the input
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 r = RX[(R_Y2 + 8 * w + 2 * g) * QS + hq], a = quad_of(acc, g);
            RX[(R_Y3 + 8 * w + 2 * g) * QS + hq] = relu4(f32x4{a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w});
        }
    }
    __syncthreads();

    STAMP(22);
    // ---- P7: s2 1x1 32 -> 32.  16 kHz: stride 2 -> column 0, wave 0.  8 kHz: stride 1 -> column w, waves 0,1 ----
    PRE_L(0)
    SB();
    if (K8 ? w < 2 : w == 0) {
        const int rin = R_Y3 + (K8 ? 8 * w : 0), rout = K8 ? R8_Y4 + 8 * w : R_Y4;
        f32x16 acc = acc_of(p7w[0], p7w[1], p7w[2], p7w[3]);
        TG_MMA(acc, p7w[4], p7w[5], p7w[6], p7w[7], RX[(rin + 0) * QS + hq], RX[(rin + 2) * QS + hq], RX[(rin + 4) * QS + hq], RX[(rin + 6) * QS + hq])
        store_tile_relu(RX, rout, m, h, acc);
    }
    __syncthreads();

    // ---- P8: block 3 (32 -> 64).  16 kHz: one column (dw: centre tap only), waves 0,1 = output tile.
    //      8 kHz: two columns, wave w = (output tile w&1, column w>>1); dw taps 2,3 on column 0 and 1,2 on column 1 ----
    if (K8 || w < 2) {
        f32x16 acc = acc_of(p8w[0], p8w[1], p8w[2], p8w[3]);
        f32x4 d[4], y[4];
        if (K8) {
            int o_ = (int)P.sect[w][S_L3];
            asm volatile("" : "+s"(o_));
            const int col = w >> 1, oth = 1 - col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 2 * j + h;
                y[j] = RX[(R8_Y4 + 8 * col + 2 * j) * QS + hq];
                const f32x4 yo = RX[(R8_Y4 + 8 * oth + 2 * j) * QS + hq];
                // out[col] = b + w[2] in[col] + w[2 + (oth - col)] in[oth]
                d[j] = relu4(fma4(ldt(wrs, q * 6 + 2 + (oth - col), o_), yo, fma4(p8t[2 * j], y[j], p8t[2 * j + 1])));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j] = RX[(R_Y4 + 2 * j) * QS + hq];
                d[j] = relu4(fma4(p8t[2 * j], y[j], p8t[2 * j + 1]));
            }
        }
        TG_MMA(acc, p8w[4], p8w[5], p8w[6], p8w[7], d[0], d[1], d[2], d[3])
        TG_MMA(acc, p8w[8], p8w[9], p8w[10], p8w[11], y[0], y[1], y[2], y[3])
        store_tile_relu(RX, K8 ? R8_Y5 + 16 * (w >> 1) + 8 * (w & 1) : R_Y5 + 8 * w, m, h, acc);
    }
    __syncthreads();

    // ---- P9: s3 1x1 64 -> 64; 16 kHz: waves 0,1 = output tile; 8 kHz: wave w = (tile w&1, column w>>1) -----------
    if (K8 || w < 2) {
        const int rin = K8 ? R8_Y5 + 16 * (w >> 1) : R_Y5;
        f32x16 acc = acc_of(p9w[0], p9w[1], p9w[2], p9w[3]);
        TG_MMA(acc, p9w[4], p9w[5], p9w[6], p9w[7], RX[(rin + 0) * QS + hq], RX[(rin + 2) * QS + hq], RX[(rin + 4) * QS + hq], RX[(rin + 6) * QS + hq])
        TG_MMA(acc, p9w[8], p9w[9], p9w[10], p9w[11], RX[(rin + 8) * QS + hq], RX[(rin + 10) * QS + hq], RX[(rin + 12) * QS + hq], RX[(rin + 14) * QS + hq])
        store_tile_relu(RX, K8 ? R8_Y6 + 16 * (w >> 1) + 8 * (w & 1) : R_Y6 + 8 * w, m, h, acc);
    }
    __syncthreads();

    STAMP(23);
    // ---- P10/P11: two stacked LSTM(64) cells, T3 time steps (1, or 2 for the 8 kHz sub-model).  Wave w owns hidden
    //      units 16w .. 16w+15 with all four gates and the full K = 128 (layer input | h_{t-1}): tile A rows = gates i|f,
    //      tile B rows = gates g|o, so i, f, g, o of a unit land in the same lane and every wave finishes its own
    //      cells - no partial sums to exchange, one barrier per cell.  Between steps h lives in LDS, c in registers.
    constexpr int T3 = K8 ? 2 : 1;
    constexpr int R_H0M = R8_Y5;                    // 8 kHz only: layer 0's h after the SECOND step (R_H0N is still being read)
#pragma unroll
    for (int step = 0; step < T3; ++step) {
    float part = 0.f;
#pragma unroll
    for (int layer = 0; layer < 2; ++layer) {
        const int ob = (int)P.sect[w][layer == 0 ? S_LSTM0 : S_LSTM1];
        f32x16 gA = acc_of(lb[0], lb[1], lb[2], lb[3]), gB = acc_of(lb[4], lb[5], lb[6], lb[7]);
        const int rx = layer == 0 ? (K8 ? R8_Y6 + 16 * step : R_Y6) : (step == 0 ? R_H0N : R_H0M);
        const int rh = layer == 0 ? (step == 0 ? R_H0 : R_H0N) : (step == 0 ? R_H1 : R_H1N);
        const f32x4 *const xsrc = RX + rx * QS + hq, *const hsrc = RX + rh * QS + hq;
        // stage s = k-iterations 2s, 2s+1 (16 MFMAs); iterations 0..7 contract the layer input, 8..15 h_{t-1}
#define LS_ROW(it) (((it) < 8 ? xsrc : hsrc)[(2 * ((it) & 7)) * QS])
#define LS_LD(S, st)                                                                              \
    S##w0 = WL(ob + 8 + 4 * (st)); S##w1 = WL(ob + 9 + 4 * (st)); S##w2 = WL(ob + 10 + 4 * (st)); S##w3 = WL(ob + 11 + 4 * (st)); \
    S##a0 = LS_ROW(2 * (st)); S##a1 = LS_ROW(2 * (st) + 1);
#define LS_MMA(S)                                                                                 \
    gA = mfma4(S##w0, S##a0, gA); gB = mfma4(S##w1, S##a0, gB); gA = mfma4(S##w2, S##a1, gA); gB = mfma4(S##w3, S##a1, gB);
        f32x4 Aw0 = lw[0], Aw1 = lw[1], Aw2 = lw[2], Aw3 = lw[3], Aa0 = LS_ROW(0), Aa1 = LS_ROW(1);
        f32x4 Bw0, Bw1, Bw2, Bw3, Ba0, Ba1;
#pragma unroll
        for (int st = 0; st < 8; st += 2) {
            LS_LD(B, st + 1) SB();
            LS_MMA(A) SB();
            const int stn = st + 2 < 8 ? st + 2 : 6;
            LS_LD(A, stn) SB();
            LS_MMA(B) SB();
        }
#undef LS_ROW
#undef LS_LD
#undef LS_MMA
        // the next cell's biases and first weights fly during this cell's update
        if (layer == 0) {
            PRE_L(1)
            SB();
        } else if (step + 1 < T3) {
            PRE_L(0)
            SB();
        }
        {
            float *st = P.state + (size_t)slot * 256;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // tile row 8 r' + 4 h + i is register 4 r' + i: rows 0..15 (r' = 0,1) = first gate, 16..31 = second gate
                const f32x4 i4 = quad_of(gA, e), f4 = quad_of(gA, 2 + e), c4g = quad_of(gB, e), o4 = quad_of(gB, 2 + e);
                const int unit = 16 * w + 8 * e + 4 * h;
                const f32x4 cp = cprev[layer][e];
                f32x4 cn, hn;
#define CELL(k)                                                             \
    cn.k = sigmoidf_(f4.k) * cp.k + sigmoidf_(i4.k) * tanhf_(c4g.k);      \
    hn.k = sigmoidf_(o4.k) * tanhf_(cn.k);
                CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
                cprev[layer][e] = cn;
                if (step == T3 - 1 && live) {
                    *reinterpret_cast<f32x4 *>(st + 128 + 64 * layer + unit) = cn;
                    *reinterpret_cast<f32x4 *>(st + 64 * layer + unit) = hn;
                }
                if (layer == 0) {
                    RX[((step == 0 ? R_H0N : R_H0M) + 4 * w + 2 * e) * QS + hq] = hn;
                } else {
                    if (step + 1 < T3) RX[(R_H1N + 4 * w + 2 * e) * QS + hq] = hn;
                    const f32x4 hw = hwq[e];
                    part += hw.x * fmaxf(hn.x, 0.f) + hw.y * fmaxf(hn.y, 0.f) + hw.z * fmaxf(hn.z, 0.f) + hw.w * fmaxf(hn.w, 0.f);
                }
            }
        }
        __syncthreads();
    }
    part += __shfl_xor(part, 32);
    if (h == 0) headp[step * 128 + w * 32 + m] = part;
    }
#undef PRE_A
#undef PRE_B
#undef PRE_C
#undef PRE_L
    STAMP(24);
    __syncthreads();

    // ---- head + state machine ------------------------------------------------------------------------
    if (sm_thread) {
        float p = sigmoidf_(hb + ((headp[tid] + headp[32 + tid]) + (headp[64 + tid] + headp[96 + tid])));
        if (K8)     // ReduceMean over the two time steps
            p = (p + sigmoidf_(hb + ((headp[128 + tid] + headp[160 + tid]) + (headp[192 + tid] + headp[224 + tid])))) * 0.5f;
        p = fminf(p, 1.0f);
        P.probs[(size_t)(tile0 + tid) * T + tframe] = p;
        SmSlot sm = smL[tid];
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

// host-callable launcher: the T frames of a call run as T launches on one stream (state lives in HBM between them)
extern "C" hipError_t vadk_launch_silero_v4(const vadk::StepParams *p, hipStream_t stream) {
    (void)hipGetLastError();   // HIP's last-error slot is sticky and process-wide: a stale failure from anywhere else must not become ours
    const int tiles = (p->n + vadk::MT - 1) / vadk::MT;
    if (tiles <= 0) return hipSuccess;
    for (int t = 0; t < p->T; ++t) {
        if (p->variant == 1)
            hipLaunchKernelGGL(silero_v4_step<true>, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p, t);
        else
            hipLaunchKernelGGL(silero_v4_step<false>, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p, t);
    }
    return hipGetLastError();
}

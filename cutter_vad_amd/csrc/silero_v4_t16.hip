// Silero-VAD V4 step kernel on 16-STREAM tiles, two workgroups per CU (gfx950).
//
// silero_v4.hip carries 32 streams per workgroup with one wave per SIMD (it needs all 512 registers to park the magnitudes
// and 158 KB of LDS), and 57 % of its cycles are waits that a single wave cannot hide: seven thin phases between barriers, the
// frame's HBM round trip, dependent LDS / L2 round trips.  This kernel is the same network and the same
// algebra (reflect-padded frame in LDS, 4-way folded DFT, the two real bins in float64, K-split first layer on 16 x 16 x 4
// tiles, LSTM waves owning all four gates of 16 units) on v_mfma_f32_16x16x4_f32 throughout: a workgroup carries 16 streams in
// under 80 KB of LDS and 256 registers per wave, so TWO workgroups share a CU and each fills the other's waits.
// Measured: 62.1 -> 50.8 us per 8 192 streams, 59.8 -> 31.6 us per 4 096 (one workgroup per CU); the MFMA pipe is
// busy 53 % of the CU-busy cycles (43 %), and the pair of workgroups issues MFMA + VALU work in ~85 % of them - what is left is
// instruction count and the ~8.5 k idle cycles at the start of every launch (kernel arguments, first half of the frame).
//
// Fragment convention as in silero_v5_t16.hip: lane l = (n = l & 15, kq = l >> 4); A: W[row n][k = kq]; B: X[k = kq][stream n];
// D: lane (stream n, rq = kq) holds rows 4 rq .. 4 rq + 3 of the 16-row tile = one LDS quad.  A k-iteration contracts 16
// channels: lane (n, kq) reads activation quad row 4 j + kq (one ds_read_b128 = the B operands of 4 MFMAs).
// LDS quad row = 16 streams x float4, dense stride QSD = 16; the folded STFT operands are written transposed by the fold (16
// lanes of a stream write 16 rows) and use the padded stride QSL = 17.
// Weight stream: pack_silero_v4_t16 (pack_weights.cpp); tests/kernel_model.py::v4_step_t16 is the NumPy model of this file's
// indexing.
#include <hip/hip_runtime.h>
#include "vad_layout.h"
#include "sm_device.h"
#include "vadk_device.h"

using namespace vadk;
using namespace vadk::dev;

// -DVADK_STAMPS (tools/kbench4.cpp -DKB_TILE16): s_memtime at phase boundaries, [block][wave][32]
#ifdef VADK_STAMPS
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (lane == 0) P.stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + (k)] = clock64();          \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

constexpr int MT16 = 16;
constexpr int QSL = 17;
constexpr int QSD = 16;
// STFT part: reflect-padded frame [16][176 quads] | folded operands of two columns [128 rows][QSL] | small
constexpr int XPQ = 176;
constexpr int U_XS = MT16 * XPQ;
constexpr int U_UV = 128 * QSL;
constexpr int U_SMALL = 40;                       // nyqv [2][16], dcv [2][16], fcor [2][3][16] floats
constexpr int K1_F4 = U_XS + U_UV + U_SMALL;
// tail: rows as in vad_layout.h v4 (33 t + q magnitudes, R_A16, R_Y*, R_H*), dense stride
constexpr int T_ROWS = vadk::v4::MAG_ROWS + 16;
constexpr int T_MISC_FLOATS = 16 + 8 * 16 + 2 * 4 * 16;      // mm [16], colmean [8][16], head partials [2 steps][4 waves][16]
constexpr int K2_F4 = T_ROWS * QSD + T_MISC_FLOATS / 4 + 128;   // + partial log sums [4 waves][8 columns][16]
constexpr int R_NY = 64;                          // tail: |X128| [8 columns][16 streams] as floats in rows 64, 65 (past the partial tiles)
constexpr int T16_LDS_F4 = K1_F4 > K2_F4 ? K1_F4 : K2_F4;
static_assert(T16_LDS_F4 * 16 <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ f32x4 ldt(__amdgpu_buffer_rsrc_t rs, int row, int blk) {   // table row (float4) of a VALU table
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, row * 16, blk * 1024, 0));
}

__device__ __forceinline__ f32x4 mfma16(f32x4 w, f32x4 a, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc, 0, 0, 0);
    return acc;
}

// Quad arithmetic as PACKED fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes of a register pair per instruction, same
// rounding as the scalar forms).  fp32 MFMAs and VALU work share the vector pipe, so every plain VALU instruction not issued is
// time; left to itself the compiler packs about a fifth of these.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 cat2(f32x2 lo, f32x2 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3); }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) {
    return cat2(__builtin_elementwise_fma(a.lo, b.lo, c.lo), __builtin_elementwise_fma(a.hi, b.hi, c.hi));
}
__device__ __forceinline__ f32x4 mul4(f32x4 a, f32x4 b) { return cat2(a.lo * b.lo, a.hi * b.hi); }
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) { return cat2(a.lo + b.lo, a.hi + b.hi); }
// a - b as one packed instruction: vadk_device.h, pk::sub2 (v_pk_fma_f32 with an opaque -1; no inline-asm instruction, which the
// hazard recogniser could not see)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return pk::sub2(a, b); }
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) { return cat2(pk_sub(a.lo, b.lo), pk_sub(a.hi, b.hi)); }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
// |re + i im| of a quad: fma(re, re, im * im) as in mag_(), packed, then the four square roots
__device__ __forceinline__ f32x4 mag4(f32x4 re, f32x4 im) {
    const f32x4 q = fma4(re, re, mul4(im, im));
    return f32x4{__builtin_amdgcn_sqrtf(q.x), __builtin_amdgcn_sqrtf(q.y), __builtin_amdgcn_sqrtf(q.z), __builtin_amdgcn_sqrtf(q.w)};
}

// a double moved across lanes by a DPP control (quad_perm / row_half_mirror) on its two halves
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// fold partner of a quad of folded samples: lane fq of a 16-lane row holds n = 4 fq + i, the result holds the values at 64 - n
// (i = 0: lane 16 - fq, component 0 = row_mirror then row_shr:1; i = 1, 2, 3: lane 15 - fq, components 3, 2, 1 = row_mirror).
// Lane 0's component 0 has no partner (n = 0 <-> 64): the DPP's bound control leaves 0 there and the caller overrides it.
__device__ __forceinline__ f32x4 mirror64(f32x4 v) {
    return f32x4{dpp_f32<0x111>(dpp_f32<0x140>(v.x)), dpp_f32<0x140>(v.w), dpp_f32<0x140>(v.z), dpp_f32<0x140>(v.y)};
}

__device__ __forceinline__ float log1p20(float mag) {   // log(1 + mag * 2^20): Mul, Add, Log of the graph
    return __builtin_amdgcn_logf(1.0f + mag * 1048576.0f) * 0.69314718055994531f;
}
__device__ __forceinline__ float lognorm(float mag, float mm) { return log1p20(mag) - mm; }     // minus the adaptive-normalisation mean
__device__ __forceinline__ f32x4 log1p20_4(f32x4 mag) {     // the same three operations on a quad, the multiply-add and the scaling packed
    const f32x4 a = fma4(mag, splat4(1048576.0f), splat4(1.0f));
    const f32x4 l = f32x4{__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(a.y), __builtin_amdgcn_logf(a.z), __builtin_amdgcn_logf(a.w)};
    return mul4(l, splat4(0.69314718055994531f));
}

}  // namespace

// K8: the graph's 8 kHz sub-model (silero_v4.hip has the differences: third stride conv has stride 1, two columns through block 3,
// two LSTM time steps, mean of the two sigmoids)
template <bool K8>
__global__ void __launch_bounds__(NTHREADS, 2) silero_v4_step16(const StepParams P, const int tframe) {
    using namespace vadk::v4;
    __shared__ f32x4 lds[T16_LDS_F4];
    f32x4 *const XP = lds;
    f32x4 *const UV = lds + U_XS;
    float *const nyqv = reinterpret_cast<float *>(UV + U_UV);    // [2][16] |X128| of the two columns in flight
    float *const dcv = nyqv + 32;                                 // [2][16] X0 (signed)
    float *const fcor = dcv + 32;                                 // [2 columns][y128, a64, b64][16 streams]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int nq = kq * QSD + n, nqL = kq * QSL + n;
    const int tile0 = blockIdx.x * MT16;
    const int T = P.T;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const int lane16 = lane * 16;
    const int o_stft = (int)P.sect[w][S_STFT];
#define WL(blk) ldw(wrs, lane16, (blk))
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // the tail's view of the same LDS
    f32x4 *const RX = lds;
    float *const misc = reinterpret_cast<float *>(lds + T_ROWS * QSD);
    float *const mmv = misc;                  // [16]
    float *const colmean = misc + 16;         // [8][16]
    float *const headp = misc + 16 + 128;     // [2][4][16]
    float *const colpart = misc + T_MISC_FLOATS;   // [4 waves][8 columns][16 streams]
    const int gf = tile0 + n;                 // tid & 15 == n: one slot lookup serves h, c and the state machine
    const bool live = gf < P.n;
    const int slot = live ? (P.slots ? P.slots[gf] : gf) : 0;

    STAMP(0);
    // window of the stored basis for this thread's fold position n = 4 fq .. 4 fq + 3 (W1 = w[n] = w[256 - n], W3 = w[128 + n] =
    // w[128 - n]: the packer checked the symmetry bit for bit): requested BEFORE the frame (a wave's loads return in order - behind the frame they would arrive with its last byte)
    const int fq = tid & 15, fms = tid >> 4;
    const int o_win = (int)P.sect[w][S_NYQ];
    const f32x4 W1 = ldw(wrs, fq * 16, o_win), W3 = ldw(wrs, (32 + fq) * 16, o_win);
    const float w64 = ldw(wrs, 16 * 16, o_win).x;                 // w[64] = w[192]
    // ... and as doubles for the float64 sums of the two real bins: converted once per frame, not once per column they serve
    const double W1d[4] = {(double)W1.x, (double)W1.y, (double)W1.z, (double)W1.w};
    const double W3d[4] = {(double)W3.x, (double)W3.y, (double)W3.z, (double)W3.w};
    const double w64d = (double)w64;
    f32x4 S0w[12];                                   // ... and so are the DFT blocks of the first column pair (6 k-iterations x {cos, -sin})
#pragma unroll
    for (int k = 0; k < 12; ++k) S0w[k] = WL(o_stft + k);
    // ---- raw frame -> LDS (gate + int16 scaling fused): 16 streams x 128 quads, 8 per thread; piece `it` = quads 16 it .. 16 it + 15
    //      of every stream.  The first column pair needs only pieces 0..3 (+ the left mirror): they are requested first, stored and
    //      folded while 4..7 - requested when 0..3 have arrived, so that HBM serves every workgroup's first half first - are on
    //      their way; those are stored under the MFMAs of column pairs 0 and 1 (-0.6 us per 8 192 streams).
    //      The buffer descriptor's range check zero-fills the streams past n.
    const float thr = P.thresh;
    const bool f32in = P.fmt == 0;
    const float isc = P.fmt == 1 ? 32767.0f : 32768.0f, risc = 1.0f / isc;
    u32x4 xv[8];
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(P.frames), 0, (int)((unsigned)P.n * (unsigned)T * (f32in ? 2048u : 1024u)), 0x00020000);
    const int xq0 = ((tile0 + fms) * T + tframe) * 128 + fq;
#define X_ISSUE(lo, hi)                                                                                     \
    if (f32in) {                                                                                            \
        _Pragma("unroll") for (int it = (lo); it < (hi); ++it)                                              \
            xv[it] = __builtin_amdgcn_raw_buffer_load_b128(frs, (xq0 + 16 * it) * 16, 0, 0);                \
    } else {                                                                                                \
        _Pragma("unroll") for (int it = (lo); it < (hi); ++it) {                                            \
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(frs, (xq0 + 16 * it) * 8, 0, 0);           \
            xv[it] = u32x4{v.x, v.y, 0u, 0u};                                                               \
        }                                                                                                   \
    }
    X_ISSUE(0, 4)
#define X_PUT(it)                                                                                           \
    {                                                                                                       \
        f32x4 v_ = __builtin_bit_cast(f32x4, xv[it]);                                                       \
        if (!f32in) {                                                                                       \
            const int s0 = (int)(short)(xv[it].x & 0xffffu), s1 = (int)(short)(xv[it].x >> 16);             \
            const int s2 = (int)(short)(xv[it].y & 0xffffu), s3 = (int)(short)(xv[it].y >> 16);             \
            v_ = f32x4{i16_div(s0, isc, risc), i16_div(s1, isc, risc), i16_div(s2, isc, risc), i16_div(s3, isc, risc)};   \
        }                                                                                                   \
        XP[fms * XPQ + 24 + 16 * (it) + fq] = gate4(v_, thr);                                               \
    }
    X_PUT(0) X_PUT(1) X_PUT(2) X_PUT(3)
    SB();
    X_ISSUE(4, 8)          // asked for only now: HBM serves every workgroup's first half first
    __syncthreads();
    // mirrored edges (numpy 'reflect', 96 + 96; silero_v4.hip has the index algebra): the left one now, the right one once pieces
    // 6, 7 are in (16 streams x 24 quads each)
#define X_MIRROR(right)                                                                                     \
    _Pragma("unroll") for (int it_ = 0; it_ < 2; ++it_) {                                                   \
        const int idx = it_ * NTHREADS + tid;                                                               \
        if (idx < MT16 * 24) {                                                                              \
            const int ms = idx / 24, k = idx - ms * 24;                                                     \
            f32x4 *row = XP + ms * XPQ;                                                                     \
            if (!(right)) {                                                                                 \
                const f32x4 lo = row[48 - k], hi = row[47 - k];                                             \
                row[k] = f32x4{lo.x, hi.w, hi.z, hi.y};                                                     \
            } else {                                                                                        \
                const int Q = 152 + k;                                                                      \
                const f32x4 p = row[303 - Q], pm = row[302 - Q];                                            \
                row[Q] = f32x4{p.z, p.y, p.x, pm.w};                                                        \
            }                                                                                               \
        }                                                                                                   \
    }
    X_MIRROR(false)
    __syncthreads();
    STAMP(1);

    f32x4 mg[8][2];                                  // |X| of this wave's 32 bins (two row tiles), 8 columns
    float nyq[4];                                    // threads < 32: |X[128]| of column 2 grp + (tid >> 4), stream n
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {              // STFT columns 2 grp, 2 grp + 1 (hop 64 on the padded frame)
        int ws = o_stft;
        asm volatile("" : "+s"(ws));
        if (grp == 2) X_MIRROR(true)      // pieces 6, 7 were stored before the barrier that ended column pair 1; read by pair 3's fold
        // ---- window + 4-way fold: stream fms, n = 4 fq + i, one column per pass (silero_v4.hip has the algebra)
        double ed[2], od[2];                         // this lane's share of E = sum over even n of w[n] x[n], O = the odd n, per column
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int Q0 = 16 * (2 * grp + cp);
            const f32x4 *xs = XP + fms * XPQ + Q0;
            const f32x4 xA = xs[fq], xC = xs[32 + fq];
            const f32x4 r1a = xs[32 - fq], r1b = xs[31 - fq];
            const f32x4 r2a = xs[fq == 0 ? 63 : 64 - fq], r2b = xs[63 - fq];
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
            // ---- the two REAL bins, k = 0 and k = 128, in float64 straight from the samples (DESIGN.md §3 "Numerics"): X0 = E + O,
            //      X128 = E - O.  The fold's own loads are the operands: this lane holds x[n], x[256 - n] (window w[n]) and x[128 - n],
            //      x[128 + n] (window w[128 + n]) for n = 4 fq + i - the window is symmetric, so each PAIR is added in float64 (exact)
            //      and multiplied once; parity of n = parity of i.  n = 0 has w = 0; n = 128 is its own partner (counted once).
            {
                const double a0 = (double)xA.x + (double)r2a.x, a1 = (double)xA.y + (double)r2b.w;
                const double a2 = (double)xA.z + (double)r2b.z, a3 = (double)xA.w + (double)r2b.y;
                const double b0 = (double)(fq == 0 ? 0.f : r1a.x) + (double)xC.x, b1 = (double)r1b.w + (double)xC.y;
                const double b2 = (double)r1b.z + (double)xC.z, b3 = (double)r1b.y + (double)xC.w;
                ed[cp] = __builtin_fma(W1d[0], a0, W3d[0] * b0) + __builtin_fma(W1d[2], a2, W3d[2] * b2);
                od[cp] = __builtin_fma(W1d[1], a1, W3d[1] * b1) + __builtin_fma(W1d[3], a3, W3d[3] * b3);
            }
            if (fq == 0) {
                pe.x = po.x = qe.x = qo.x = 0.f;
                const float x64 = xs[16].x, x192 = xs[48].x;
                const float y64 = x64 * w64, y192 = x192 * w64;
                fcor[(cp * 3 + 0) * 16 + fms] = y3.x;           // y[128]
                fcor[(cp * 3 + 1) * 16 + fms] = y64 + y192;     // a64
                fcor[(cp * 3 + 2) * 16 + fms] = y64 - y192;     // b64
                ed[cp] = __builtin_fma(w64d, (double)x64 + (double)x192, ed[cp]);      // n = 64, 192: even, unpaired by the fold
            }
            // the even bins k = 2 m fold once more, about n = 32 (vad_layout.h, bin_of_channel_t16): m even pairs pe[n] + pe[64 - n]
            // and qe[n] - qe[64 - n], m odd the opposite signs, n = 1..31; the unpaired n = 32 (lane 8, component 0) rides in the
            // free slot n = 0: pe[32] for m even, qe[32] for m odd (the packer put cos / -sin of n = 32 there), zero in the other two
            const f32x4 pm = mirror64(pe), qm = mirror64(qe);
            f32x4 pee = add4(pe, pm), peo = sub4(pe, pm), qee = sub4(qe, qm), qeo = add4(qe, qm);
            const float pe32 = dpp_f32<0x108>(pe.x), qe32 = dpp_f32<0x108>(qe.x);      // row_shl:8: lane 0 <- lane 8
            if (fq == 0) { pee.x = pe32; peo.x = 0.f; qee.x = 0.f; qeo.x = qe32; }
            // rows of a column: po 0..15 | qo 16..31 | pee 32..39 | peo 40..47 | qee 48..55 | qeo 56..63
            st2(&UV[(64 * cp + fq) * QSL + fms], po);
            st2(&UV[(64 * cp + 16 + fq) * QSL + fms], qo);
            if (fq < 8) {
                st2(&UV[(64 * cp + 32 + fq) * QSL + fms], pee);
                st2(&UV[(64 * cp + 40 + fq) * QSL + fms], peo);
                st2(&UV[(64 * cp + 48 + fq) * QSL + fms], qee);
                st2(&UV[(64 * cp + 56 + fq) * QSL + fms], qeo);
            }
        }
        // ---- E, O of both columns summed over the stream's 16 lanes.  Four doubles per lane; instead of four butterflies of four
        //      values each, every step HALVES what a lane carries: step 1 (lane ^ 8) leaves lanes 0..7 with column 0 and lanes 8..15
        //      with column 1, step 2 (7 - lane within a half row) leaves the lower four lanes of a half with E and the upper four
        //      with O, steps 3 and 4 finish inside a quad; then E and O meet once more.  27 instead of 48 float64-rate instructions.
        {
            const bool up8 = (fq & 8) != 0, up4 = (fq & 4) != 0;
            const double se = up8 ? ed[0] : ed[1], so = up8 ? od[0] : od[1];       // what the lane gives away
            double ke = up8 ? ed[1] : ed[0], ko = up8 ? od[1] : od[0];             // what it keeps
            ke += dpp_f64<0x128>(se); ko += dpp_f64<0x128>(so);                     // row_ror:8
            double v = (up4 ? ko : ke) + dpp_f64<0x141>(up4 ? ke : ko);             // row_half_mirror
            v += dpp_f64<0xB1>(v);                                                  // quad_perm [1,0,3,2]
            v += dpp_f64<0x4E>(v);                                                  // quad_perm [2,3,0,1]
            const double other = dpp_f64<0x141>(v);                                 // lanes 0..3 of a half: E here, O there
            if ((fq & 7) == 0) {
                dcv[(fq >> 3) * 16 + fms] = (float)(v + other);
                nyqv[(fq >> 3) * 16 + fms] = fabsf((float)(v - other));
            }
        }
        f32x4 Aw[2], Bw[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) Aw[k] = grp == 0 ? S0w[k] : WL(ws + k);
        SB();
        if (grp == 0) STAMP(2);
        __syncthreads();
        if (grp == 0) STAMP(3);
        // ---- MFMA: wave w, row tile 0 = its 16 odd bins: cos on po, -sin on qo, K = 64 (k-iterations 0..3); row tile 1 = its 16
        //      even bins: cos on pee | peo, -sin on qee | qeo (waves 2, 3 | waves 0, 1), K = 32 (k-iterations 4, 5); two columns;
        //      the accumulators start from the rank-1 terms of n = 0, 64, 128 (register i of a D quad is tile row 4 rq + i):
        //      odd k: re = -y128, im = -+ b64 (sin(pi k / 2) alternates along the rows); even k = 2 m: re = y128 + (-1)^m a64, im = 0
        f32x4 are[2][2], aim[2][2];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const float y128 = fcor[(cp * 3 + 0) * 16 + n], a64 = fcor[(cp * 3 + 1) * 16 + n], b64 = fcor[(cp * 3 + 2) * 16 + n];
            const float re1 = w < 2 ? y128 - a64 : y128 + a64;
            are[cp][0] = f32x4{-y128, -y128, -y128, -y128};
            aim[cp][0] = f32x4{-b64, b64, -b64, b64};
            are[cp][1] = f32x4{re1, re1, re1, re1};
            aim[cp][1] = zero4;
        }
        {
            const int eR = w < 2 ? 40 : 32, eI = w < 2 ? 56 : 48;
            const f32x4 *const XB = UV + nqL;
            // operand rows of k-iteration t: odd tile t = 0..3: po rows 4 t + kq, qo rows 16 + 4 t + kq; even tile t = 4, 5
#define S_ROW_R(t) ((t) < 4 ? 4 * (t) : eR + 4 * ((t) - 4))
#define S_ROW_I(t) ((t) < 4 ? 16 + 4 * (t) : eI + 4 * ((t) - 4))
            f32x4 Au[2], Av[2], Bu[2], Bv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { Au[c] = XB[(64 * c + S_ROW_R(0)) * QSL]; Av[c] = XB[(64 * c + S_ROW_I(0)) * QSL]; }
#define S_LD(S, tt)                                                                        \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) S##w[k] = grp == 0 ? S0w[2 * (tt) + k] : WL(ws + 2 * (tt) + k);   \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) { S##u[c] = XB[(64 * c + S_ROW_R(tt)) * QSL]; S##v[c] = XB[(64 * c + S_ROW_I(tt)) * QSL]; }
#define S_MMA(S, rt)                                                                       \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                        \
        are[c][rt] = mfma16(S##w[0], S##u[c], are[c][rt]);                                 \
        aim[c][rt] = mfma16(S##w[1], S##v[c], aim[c][rt]);                                 \
    }
            S_LD(B, 1) SB();
            S_MMA(A, 0) SB();
            S_LD(A, 2) SB();
            S_MMA(B, 0) SB();
            S_LD(B, 3) SB();
            S_MMA(A, 0) SB();
            S_LD(A, 4) SB();
            S_MMA(B, 0) SB();
            S_LD(B, 5) SB();
            S_MMA(A, 1) SB();
            S_MMA(B, 1) SB();
#undef S_LD
#undef S_MMA
#undef S_ROW_R
#undef S_ROW_I
        }
        if (grp == 0) STAMP(4);
        {   // bin 0 (wave 2, row tile 1, tile row 0 = component 0 of the lanes kq = 0) takes the float64 sum; its im is identically 0
            const bool own0 = (w == 2) & (kq == 0);
            const float d0 = dcv[n], d1 = dcv[16 + n];
            are[0][1].x = own0 ? d0 : are[0][1].x;
            are[1][1].x = own0 ? d1 : are[1][1].x;
        }
#pragma unroll
        for (int cp = 0; cp < 2; ++cp)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 r = are[cp][rt], i = aim[cp][rt];
                mg[2 * grp + cp][rt] = mag4(r, i);
            }
        nyq[grp] = nyqv[((tid >> 4) & 1) * 16 + n];
        if (grp == 0) { X_PUT(4) X_PUT(5) }
        if (grp == 1) { X_PUT(6) X_PUT(7) }
        if (grp == 0) STAMP(5);
        __syncthreads();       // every wave done with UV / fcor / dcv before the next fold overwrites them
        if (grp == 0) STAMP(6);
        if (grp == 3) STAMP(7);
    }

#undef X_PUT
#undef X_ISSUE
#undef X_MIRROR
    // =================================================================================================
    //  tail: the LDS is re-used with the second layout from here on
    // =================================================================================================
    // state of this lane's stream: h_{t-1} of both layers -> rows R_H0.. (32 quads per stream; nothing else lives there in this
    // kernel's tail), c_{t-1} of this lane's units -> registers
    f32x4 cprev[2];
    {
        const float *st = P.state + (size_t)slot * 256;
        f32x4 hprev[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(st)[(tid >> 4) * 2 + qq];
            hprev[qq] = live ? v : zero4;
        }
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(st + 128 + 64 * layer + 16 * w + 4 * kq);   // c of units 16 w + 4 kq + i
            cprev[layer] = live ? v : zero4;
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) RX[(R_H0 + (tid >> 4) * 2 + qq) * QSD + (tid & 15)] = hprev[qq];
    }
    // ---- P0 + P1: the log-spectrum of this wave's bins (kept in registers next to the magnitudes: the first layer below contracts
    //      exactly the channel quads a wave produced, so neither ever travels through LDS) and its per-column means
    f32x4 lg[8][2];
#pragma unroll
    for (int tc = 0; tc < 8; ++tc) {
        float s = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const f32x4 v = mg[tc][rt];
            const f32x4 l = log1p20_4(v);
            lg[tc][rt] = l;
            s += (l.x + l.y) + (l.z + l.w);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (kq == 0) colpart[(w * 8 + tc) * 16 + n] = s;
    }
    float *const nyqm = reinterpret_cast<float *>(RX + R_NY * QSD);     // [8 columns][16 streams] |X128|
    if (tid < 32) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) nyqm[(2 * grp + (tid >> 4)) * 16 + n] = nyq[grp];
    }
    __syncthreads();
    if (tid < 128) {
        const int ms = tid & 15, tc = tid >> 4;
        const float s = ((colpart[tc * 16 + ms] + colpart[(8 + tc) * 16 + ms]) + (colpart[(16 + tc) * 16 + ms] + colpart[(24 + tc) * 16 + ms])) +
                        log1p20(nyqm[tc * 16 + ms]);
        colmean[tc * 16 + ms] = s * (1.0f / 129.0f);
    }
    __syncthreads();
    if (tid < MT16) {
        const int o_dw0 = (int)P.sect[0][S_DW0];
        const f32x4 f0 = ldt(wrs, 2 * 34 * 6, o_dw0), f1 = ldt(wrs, 2 * 34 * 6 + 1, o_dw0);
        const float filt[7] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z};
        float mp[14];
#pragma unroll
        for (int t = 0; t < 8; ++t) mp[3 + t] = colmean[t * 16 + tid];
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
    STAMP(17);

    // ---- P2: first layer (258 -> 16 channels), all four kept output columns in every wave, K split over the waves: wave w
    //      contracts the k-iterations j = 2 w, 2 w + 1 - the 32 bins it produced itself, straight from its registers; the Nyquist
    //      channel is a rank-1 VALU term (output column w in wave w); the four partial tiles per column meet in LDS (silero_v4.hip, P2)
    f32x4 p3b, p3w;
    {
        int o_dw0 = (int)P.sect[w][S_DW0], o_l0 = (int)P.sect[w][S_L0];
        asm volatile("" : "+s"(o_dw0), "+s"(o_l0));
        f32x4 acc[4];
        {
            const f32x4 b0 = WL(o_l0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = w == 0 ? b0 : zero4;       // the bias enters once
        }
        const int ws = o_l0 + 5;
        const float mm = mmv[n];
        f32x4 tn[12], wn[4];         // Nyquist channel: quad 32 of the tables (component 0), its four weight columns in the D layout
        tn[0] = ldt(wrs, 32 * 6 + 5, o_dw0); tn[1] = ldt(wrs, (34 + 32) * 6 + 5, o_dw0);
#pragma unroll
        for (int k = 0; k < 5; ++k) { tn[2 + 2 * k] = ldt(wrs, 32 * 6 + k, o_dw0); tn[3 + 2 * k] = ldt(wrs, (34 + 32) * 6 + k, o_dw0); }
#pragma unroll
        for (int k = 0; k < 4; ++k) wn[k] = WL(o_l0 + 1 + k);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q_ = 4 * (2 * w + it) + kq;
            f32x4 tb[12], wq[4];
            tb[0] = ldt(wrs, q_ * 6 + 5, o_dw0); tb[1] = ldt(wrs, (34 + q_) * 6 + 5, o_dw0);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                tb[2 + 2 * k] = ldt(wrs, q_ * 6 + k, o_dw0);
                tb[3 + 2 * k] = ldt(wrs, (34 + q_) * 6 + k, o_dw0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) wq[k] = WL(ws + 4 * (2 * w + it) + k);
            f32x4 mgc[8], spc[8];
#pragma unroll
            for (int tc = 0; tc < 8; ++tc) {
                mgc[tc] = mg[tc][it];
                const f32x4 l = lg[tc][it];
                spc[tc] = sub4(l, splat4(mm));       // Sub of the graph: the adaptive normalisation
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 dm = tb[0], dn = tb[1];        // depthwise k5 p2 around input column 2c: biases, then the taps
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int tc = 2 * c + k - 2;
                    if (tc >= 0 && tc < 8) {
                        dm = fma4(tb[2 + 2 * k], mgc[tc], dm);
                        dn = fma4(tb[3 + 2 * k], spc[tc], dn);
                    }
                }
                dm = relu4(dm);
                dn = relu4(dn);
                f32x4 a_ = acc[c];
                a_ = mfma16(wq[0], dm, a_);
                a_ = mfma16(wq[1], mgc[2 * c], a_);
                a_ = mfma16(wq[2], dn, a_);
                a_ = mfma16(wq[3], spc[2 * c], a_);
                acc[c] = a_;
            }
        }
        {   // Nyquist channel of output column w (input column 2w): scalars per stream, rank-1 into this lane's output quad
            float dm = tn[0].x, dn = tn[1].x, xm = 0.f, xn = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = 2 * w + k - 2;
                if (tc >= 0 && tc < 8) {             // wave-uniform
                    const float mgv = nyqm[tc * 16 + n];
                    const float sp = lognorm(mgv, mm);
                    dm = fmaf(tn[2 + 2 * k].x, mgv, dm);
                    dn = fmaf(tn[3 + 2 * k].x, sp, dn);
                    if (k == 2) { xm = mgv; xn = sp; }
                }
            }
            dm = fmaxf(dm, 0.f);
            dn = fmaxf(dn, 0.f);
            const f32x4 r1 = wn[0] * dm + wn[1] * xm + wn[2] * dn + wn[3] * xn;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c == w) acc[c] = acc[c] + r1;
        }
        {
            const int o_ = (int)P.sect[w][S_S0];
            p3b = WL(o_); p3w = WL(o_ + 1);
        }
        STAMP(18);
#pragma unroll
        for (int c = 0; c < 4; ++c) RX[((w * 4 + c) * 4) * QSD + nq] = acc[c];       // PART[wave][column][output quad][stream], rows 0..63
        __syncthreads();
        {
            const int cq = tid >> 4, ms = tid & 15;   // (column, channel quad), stream
            const f32x4 p0 = RX[cq * QSD + ms], p1 = RX[(16 + cq) * QSD + ms], p2 = RX[(32 + cq) * QSD + ms], p3 = RX[(48 + cq) * QSD + ms];
            RX[(R_A16 + cq) * QSD + ms] = relu4((p0 + p1) + (p2 + p3));
        }
    }
    __syncthreads();
    STAMP(20);

    // ---- P3: s0 1x1 16 -> 16 on the 4 kept columns; wave w = column w -------------------------------------------------------
    f32x4 p4t[6], p4w[6];
    {
        const int o_ = (int)P.sect[w][S_L1];
#pragma unroll
        for (int k = 0; k < 6; ++k) p4t[k] = ldt(wrs, kq * 6 + k, o_);
#pragma unroll
        for (int k = 0; k < 6; ++k) p4w[k] = WL(o_ + 1 + k);
        SB();
        const f32x4 acc = mfma16(p3w, RX[(R_A16 + 4 * w) * QSD + nq], p3b);
        RX[(R_Y0 + 4 * w) * QSD + nq] = relu4(acc);
    }
    __syncthreads();

    // ---- P4: block 1 (16 -> 32): dw k5 over the 4 columns (VALU) -> pw, + proj(y); wave w = column w, two row tiles ------------
    f32x4 p5w[3];
    {
        f32x4 d = p4t[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int tc = w + k - 2;
            if (tc >= 0 && tc < 4) d = fma4(p4t[k], RX[(R_Y0 + 4 * tc) * QSD + nq], d);
        }
        d = relu4(d);
        const f32x4 y = RX[(R_Y0 + 4 * w) * QSD + nq];
        {
            const int o_ = (int)P.sect[w][S_S1] + 3 * (w & 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) p5w[k] = WL(o_ + k);
        }
        SB();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 acc = p4w[3 * rt];
            acc = mfma16(p4w[3 * rt + 1], d, acc);
            acc = mfma16(p4w[3 * rt + 2], y, acc);
            RX[(R_Y1 + 8 * w + 4 * rt) * QSD + nq] = relu4(acc);
        }
    }
    __syncthreads();

    STAMP(21);
    // ---- P5: s1 1x1 32 -> 32, stride 2: columns 0 and 2; wave w = (column w >> 1, row tile w & 1) -------------------------------
    const int col2 = w >> 1, rt2 = w & 1;
    f32x4 p6t[12], p6w[3];
    {
        const int o_ = (int)P.sect[w][S_L2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 6; ++k) p6t[6 * j + k] = ldt(wrs, (4 * j + kq) * 6 + k, o_);
#pragma unroll
        for (int k = 0; k < 3; ++k) p6w[k] = WL(o_ + 1 + 3 * rt2 + k);
        SB();
        const int r = R_Y1 + 8 * (2 * col2);
        f32x4 acc = p5w[0];
        acc = mfma16(p5w[1], RX[(r + 0) * QSD + nq], acc);
        acc = mfma16(p5w[2], RX[(r + 4) * QSD + nq], acc);
        RX[(R_Y2 + 8 * col2 + 4 * rt2) * QSD + nq] = relu4(acc);
    }
    __syncthreads();

    // ---- P6: block 2 (32 -> 32, identity residual) on 2 columns; wave w = (column, row tile) -------------------------------------
    f32x4 p7w[3];
    {
        f32x4 d[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            d[j] = p6t[6 * j + 5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int tc = col2 + k - 2;
                if (tc >= 0 && tc < 2) d[j] = fma4(p6t[6 * j + k], RX[(R_Y2 + 8 * tc + 4 * j) * QSD + nq], d[j]);
            }
            d[j] = relu4(d[j]);
        }
        {
            const int o_ = (int)P.sect[w][S_S2] + 3 * (K8 ? rt2 : (w & 1));
#pragma unroll
            for (int k = 0; k < 3; ++k) p7w[k] = WL(o_ + k);
        }
        SB();
        f32x4 acc = p6w[0];
        acc = mfma16(p6w[1], d[0], acc);
        acc = mfma16(p6w[2], d[1], acc);
        // + identity residual: output channels 16 rt + 4 kq + i = input quad 4 rt + kq of the same column
        const f32x4 r = RX[(R_Y2 + 8 * col2 + 4 * rt2) * QSD + nq];
        RX[(R_Y3 + 8 * col2 + 4 * rt2) * QSD + nq] = relu4(acc + r);
    }
    __syncthreads();

    STAMP(22);
    // ---- P7: s2 1x1 32 -> 32.  16 kHz: stride 2 -> column 0, waves 0, 1 = row tile.  8 kHz: stride 1, wave w = (column, row tile) ----
    f32x4 p8t[6], p8w[5];
    {
        const int o_ = (int)P.sect[w][S_L3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            p8t[3 * j] = ldt(wrs, (4 * j + kq) * 6 + 2, o_);
            p8t[3 * j + 1] = ldt(wrs, (4 * j + kq) * 6 + 5, o_);
            p8t[3 * j + 2] = zero4;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) p8w[k] = WL(o_ + 1 + 5 * w + k);
        SB();
        if (K8 || w < 2) {
            const int c7 = K8 ? col2 : 0, r7 = K8 ? rt2 : w;
            const int rin = R_Y3 + 8 * c7, rout = K8 ? R8_Y4 + 8 * c7 : R_Y4;
            f32x4 acc = p7w[0];
            acc = mfma16(p7w[1], RX[(rin + 0) * QSD + nq], acc);
            acc = mfma16(p7w[2], RX[(rin + 4) * QSD + nq], acc);
            RX[(rout + 4 * r7) * QSD + nq] = relu4(acc);
        }
    }
    __syncthreads();

    // ---- P8: block 3 (32 -> 64); wave w = row tile w.  16 kHz: one column (dw: centre tap only).  8 kHz: two columns (dw taps
    //      2,3 on column 0 and 1,2 on column 1) ------------------------------------------------------------------------------------
    constexpr int NC = K8 ? 2 : 1;
    f32x4 p9w[5];
    {
        {
            const int o_ = (int)P.sect[w][S_S3] + 5 * w;
#pragma unroll
            for (int k = 0; k < 5; ++k) p9w[k] = WL(o_ + k);
        }
        f32x4 y[NC][2], d[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) y[c][j] = RX[((K8 ? R8_Y4 + 8 * c : R_Y4) + 4 * j) * QSD + nq];
        if constexpr (K8) {
            const int o_ = (int)P.sect[w][S_L3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 t1 = ldt(wrs, (4 * j + kq) * 6 + 1, o_), t3 = ldt(wrs, (4 * j + kq) * 6 + 3, o_);
                // out[col] = b + w[2] in[col] + w[2 + (oth - col)] in[oth]
                d[0][j] = relu4(fma4(t3, y[1][j], fma4(p8t[3 * j], y[0][j], p8t[3 * j + 1])));
                d[1][j] = relu4(fma4(t1, y[0][j], fma4(p8t[3 * j], y[1][j], p8t[3 * j + 1])));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) d[0][j] = relu4(fma4(p8t[3 * j], y[0][j], p8t[3 * j + 1]));
        }
        SB();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 acc = p8w[0];
            acc = mfma16(p8w[1], d[c][0], acc);
            acc = mfma16(p8w[2], d[c][1], acc);
            acc = mfma16(p8w[3], y[c][0], acc);
            acc = mfma16(p8w[4], y[c][1], acc);
            RX[((K8 ? R8_Y5 + 16 * c : R_Y5) + 4 * w) * QSD + nq] = relu4(acc);
        }
    }
    __syncthreads();

    // ---- P9: s3 1x1 64 -> 64; wave w = row tile w (8 kHz: both columns) ----------------------------------------------------------
    f32x4 smq[6];
    const float hb = P.wstream[(size_t)P.sect[0][S_HEADB] * BLK_FLOATS];
    const f32x4 hw = WL((int)P.sect[w][S_HEADB] + 1 + w);
    f32x4 lbq[4];                                    // gate biases of the first cell
    {
        const int ob = (int)P.sect[w][S_LSTM0];
#pragma unroll
        for (int k = 0; k < 4; ++k) lbq[k] = WL(ob + k);
#pragma unroll
        for (int k = 0; k < 6; ++k) smq[k] = reinterpret_cast<const f32x4 *>(P.sm + slot)[k];
        SB();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int rin = K8 ? R8_Y5 + 16 * c : R_Y5;
            f32x4 acc = p9w[0];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma16(p9w[1 + j], RX[(rin + 4 * j) * QSD + nq], acc);
            RX[((K8 ? R8_Y6 + 16 * c : R_Y6) + 4 * w) * QSD + nq] = relu4(acc);
        }
    }
    __syncthreads();

    STAMP(23);
    // ---- P10/P11: two stacked LSTM(64) cells, T3 time steps (1, or 2 for the 8 kHz sub-model).  Wave w owns hidden units
    //      16 w .. 16 w + 15: four 16-row tiles = gates i, f, g, o, full K = 128 (layer input | h_{t-1}) = 8 k-iterations; the D
    //      layout puts the four gates of a unit into the same lane - every wave finishes its own cells, one barrier per cell.
    constexpr int T3 = K8 ? 2 : 1;
    constexpr int R_H0M = R8_Y5;                    // 8 kHz only: layer 0's h after the SECOND step (R_H0N is still being read)
#pragma unroll
    for (int step = 0; step < T3; ++step) {
        float part = 0.f;
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            const int ob = (int)P.sect[w][layer == 0 ? S_LSTM0 : S_LSTM1];
            f32x4 G[4] = {lbq[0], lbq[1], lbq[2], lbq[3]};
            const int rx = layer == 0 ? (K8 ? R8_Y6 + 16 * step : R_Y6) : (step == 0 ? R_H0N : R_H0M);
            const int rh = layer == 0 ? (step == 0 ? R_H0 : R_H0N) : (step == 0 ? R_H1 : R_H1N);
            const f32x4 *const xsrc = RX + rx * QSD + nq, *const hsrc = RX + rh * QSD + nq;
#define LS_ROW(it) (((it) < 4 ? xsrc : hsrc)[(4 * ((it) & 3)) * QSD])
#define LS_LD(S, it) _Pragma("unroll") for (int k = 0; k < 4; ++k) S##w[k] = WL(ob + 4 + 4 * (it) + k); S##a = LS_ROW(it);
#define LS_MMA(S) _Pragma("unroll") for (int k = 0; k < 4; ++k) G[k] = mfma16(S##w[k], S##a, G[k]);
            f32x4 Aw[4], Bw[4], Aa, Ba;
            LS_LD(A, 0)
#pragma unroll
            for (int it = 0; it < 8; it += 2) {
                LS_LD(B, it + 1) SB();
                LS_MMA(A) SB();
                const int itn = it + 2 < 8 ? it + 2 : 6;
                LS_LD(A, itn) SB();
                LS_MMA(B) SB();
            }
#undef LS_ROW
#undef LS_LD
#undef LS_MMA
            // the next cell's biases fly during this cell's update
            if (layer == 0 || step + 1 < T3) {
                const int obn = (int)P.sect[w][layer == 0 ? S_LSTM1 : S_LSTM0];
#pragma unroll
                for (int k = 0; k < 4; ++k) lbq[k] = WL(obn + k);
                SB();
            }
            {
                float *st = P.state + (size_t)slot * 256;
                const f32x4 i4 = G[0], f4 = G[1], c4g = G[2], o4 = G[3];
                const int unit = 16 * w + 4 * kq;
                const f32x4 cp = cprev[layer];
                f32x4 cn, hn;
#define CELL(k)                                                             \
    cn.k = sigmoidf_(f4.k) * cp.k + sigmoidf_(i4.k) * tanhf_(c4g.k);      \
    hn.k = sigmoidf_(o4.k) * tanhf_(cn.k);
                CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
                cprev[layer] = cn;
                if (step == T3 - 1 && live) {
                    *reinterpret_cast<f32x4 *>(st + 128 + 64 * layer + unit) = cn;
                    *reinterpret_cast<f32x4 *>(st + 64 * layer + unit) = hn;
                }
                if (layer == 0) {
                    RX[((step == 0 ? R_H0N : R_H0M) + 4 * w) * QSD + nq] = hn;
                } else {
                    if (step + 1 < T3) RX[(R_H1N + 4 * w) * QSD + nq] = hn;
                    part += hw.x * fmaxf(hn.x, 0.f) + hw.y * fmaxf(hn.y, 0.f) + hw.z * fmaxf(hn.z, 0.f) + hw.w * fmaxf(hn.w, 0.f);
                }
            }
            __syncthreads();
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (kq == 0) headp[step * 64 + w * 16 + n] = part;
    }
    STAMP(24);
    __syncthreads();

    // ---- head + state machine ------------------------------------------------------------------------
    if (tid < MT16 && live) {
        float p = sigmoidf_(hb + ((headp[tid] + headp[16 + tid]) + (headp[32 + tid] + headp[48 + tid])));
        if (K8)     // ReduceMean over the two time steps
            p = (p + sigmoidf_(hb + ((headp[64 + tid] + headp[80 + tid]) + (headp[96 + tid] + headp[112 + tid])))) * 0.5f;
        p = fminf(p, 1.0f);
        P.probs[(size_t)(tile0 + tid) * T + tframe] = p;
        SmSlot sm;
#pragma unroll
        for (int k = 0; k < 6; ++k) reinterpret_cast<f32x4 *>(&sm)[k] = smq[k];
        int seg = 0;
        const int ev = sm_step(sm, p, &seg);
        P.sm[slot] = sm;
        if (P.events) P.events[(size_t)(tile0 + tid) * T + tframe] = (uint8_t)ev;
        if (P.seg_frames) {
            if (ev & 2) P.seg_frames[tile0 + tid] = seg;
            else if (tframe == 0) P.seg_frames[tile0 + tid] = 0;
        }
    }
    STAMP(25);
#undef WL
}

// host-callable launcher: the T frames of a call run as T launches on one stream (state lives in HBM between them).
// one_per_cu: pad the launch with dynamic LDS so that a CU takes ONE workgroup - a call with at most one tile per CU spreads over
// twice as many CUs instead of pairing its tiles up (the dispatcher fills a CU before it moves on)
extern "C" hipError_t vadk_launch_silero_v4_t16(const vadk::StepParams *p, int one_per_cu, hipStream_t stream) {
    (void)hipGetLastError();
    const int tiles = (p->n + MT16 - 1) / MT16;
    if (tiles <= 0) return hipSuccess;
    const unsigned pad = one_per_cu ? 8u * 1024u : 0u;
    for (int t = 0; t < p->T; ++t) {
        if (p->variant == 1)
            hipLaunchKernelGGL(silero_v4_step16<true>, dim3(tiles), dim3(vadk::NTHREADS), pad, stream, *p, t);
        else
            hipLaunchKernelGGL(silero_v4_step16<false>, dim3(tiles), dim3(vadk::NTHREADS), pad, stream, *p, t);
    }
    return hipGetLastError();
}

// Batched Fourier resampler to 16 kHz for MI355X / gfx950.
//
// Replaces AudioUtils.resample_audio (/root/reference/src/real_time_vad/utils/audio.py:19-55 ->
// scipy.signal.resample, Fourier method, window=None) for chunks that yield exactly 512 output
// samples: 256 (8 kHz) / 768 (24 kHz) / 1536 (48 kHz) input samples per chunk.
//
// For a fixed (n_in, 512) the Fourier method is a fixed linear operator R[512][n_in]
// (SURVEY §8 a11); the host builds it in double precision from the closed form of scipy's
// spectrum copy (pack_weights.cpp: build_resample_operator) and packs it like every other weight
// stream.  The kernel is the same MFMA skeleton as the model kernels: one workgroup = 32 streams,
// weights (operator rows) on the A operand, the stream tile on the B operand, v_mfma_f32_32x32x2_f32.
// The input is folded about its midpoint on the way into LDS (chunks of 128 folded samples, double
// buffered), which halves the contraction length - see the kernel's comment.
#include <hip/hip_runtime.h>
#include "vad_layout.h"

using namespace vadk;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

namespace {

__device__ __forceinline__ f32x16 mfma4(f32x4 w, f32x4 a, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a.w, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x4 ldw(__amdgpu_buffer_rsrc_t rs, int voff, int blk) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, blk * 1024, 0));
}

__device__ __forceinline__ f32x4 quad_of(const f32x16 &a, int g) {
    switch (g) {
        case 0: return f32x4{a.s0, a.s1, a.s2, a.s3};
        case 1: return f32x4{a.s4, a.s5, a.s6, a.s7};
        case 2: return f32x4{a.s8, a.s9, a.sa, a.sb};
        default: return f32x4{a.sc, a.sd, a.se, a.sf};
    }
}

}  // namespace

// The operator has a half-period shift symmetry and a mirror symmetry (pack_weights.cpp: pack_resample_operator spells
// out the algebra), so the kernel contracts four folded inputs of length Q = n_in / 4
//   ue / ve = (x[j] + x[j+H]) +/- (x[H-j] + x[n-j]),   uo / vo = (x[j] - x[j+H]) +/- (x[H-j] - x[n-j])      (H = n_in / 2)
// against four 128-row operators (se, ae, so, ao) and recombines
//   y[o] = se+ae+so+ao, y[o+256] = se+ae-so-ao, y[256-o] = se-ae+so-ao, y[512-o] = se-ae-so+ao   (o < 128; 128, 384 on the VALU)
// - a quarter of the dense product's MFMAs.
// NT = 2: one workgroup per chunk tile, wave w = row tile w (o = 32w..32w+31) with all four parts.
// NT = 1: two workgroups (blockIdx.y) share a chunk tile: wave (y, w) = row tile 2y + (w >> 1), parts (se, ae) or (so, ao)
// by w & 1, partner waves swap their sums through LDS at the end - used when the call has too few chunk tiles to fill the
// 256 CUs (the input is read twice, from L2).
template <int NT>
__global__ void __launch_bounds__(NTHREADS, 1) vadk_resample_512(const vadk::ResampleParams PP) {
    constexpr int CH_ROWS = 16;                       // one K chunk = 64 folded samples = 16 quad rows of each of ue, ve, uo, vo
    constexpr int NP = 2 * NT;                        // accumulators per wave
    constexpr int BUF = 4 * CH_ROWS * QS;
    __shared__ f32x4 lds[2 * BUF + 64];
    float *const red = reinterpret_cast<float *>(lds + 2 * BUF);                    // [8 parts][32 streams]: rows 128 / 384
    // which segment does this workgroup serve?  (block-uniform: scalar compares on kernel arguments)
    int sidx = 0;
#pragma unroll
    for (int k = 1; k < RESAMPLE_MAX_SEGS; ++k)
        if (k < PP.nseg && (int)blockIdx.x >= PP.tile_start[k]) sidx = k;
    const vadk::ResampleSeg P = PP.seg[sidx];
    const int tile_in_seg = (int)blockIdx.x - PP.tile_start[sidx];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int hq = h * QS + m;
    const int tile0 = tile_in_seg * MT;
    const int Q = P.n_in >> 2;                        // folded length in samples = quads per input chunk
    const int nchunks = Q >> 6;
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.in), 0, (int)((unsigned)P.n * (unsigned)P.n_in * 4u), 0x00020000);
    const int lane16 = lane * 16;
    const int rt = NT == 2 ? w : 2 * (int)blockIdx.y + (w >> 1);     // this wave's 32-row output tile (0..3)
    const int po = NT == 2 ? 0 : (w & 1);                            // NT = 1: 0 = the (se, ae) pair, 1 = (so, ao)
    const int wbase = rt * (int)P.tile_blocks;

    // the one sample each half-size product cannot pair, x[Q] +/- x[Q + H], enters as a rank-1 term: accumulator init
    f32x16 acc[NP];
    {
        const int g2 = tile0 + m;
        const float xa_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (g2 * P.n_in + Q) * 4, 0, 0));
        const float xb_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (g2 * P.n_in + 3 * Q) * 4, 0, 0));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tb = wbase + 4 * (NT == 2 ? t : po);
            const float xmid = (NT == 2 ? t : po) == 0 ? xa_ + xb_ : xa_ - xb_;
            const f32x4 r0 = ldw(wrs, lane16, tb), r1 = ldw(wrs, lane16, tb + 1), r2 = ldw(wrs, lane16, tb + 2), r3 = ldw(wrs, lane16, tb + 3);
            acc[2 * t] = f32x16{r0.x * xmid, r0.y * xmid, r0.z * xmid, r0.w * xmid, r1.x * xmid, r1.y * xmid, r1.z * xmid, r1.w * xmid,
                                r2.x * xmid, r2.y * xmid, r2.z * xmid, r2.w * xmid, r3.x * xmid, r3.y * xmid, r3.z * xmid, r3.w * xmid};
            acc[2 * t + 1] = (f32x16)(0.f);
        }
    }

    // chunk loader: 32 streams x 16 folded quads, 2 per thread.  Folded quad q (j = 4q..4q+3) needs x[j] (quad q), x[j+H]
    // (quad q + Q/2), x[H-j] (quad Q/2 - q element 0, quad Q/2 - q - 1 elements 3, 2, 1) and x[n-j] (likewise from Q - q)
    u32x4 xl[2][6];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * NTHREADS + tid;
            const int ms = idx >> 4, q = (idx & 15) + 16 * c;
            const int base = (tile0 + ms) * Q;                       // streams past n: out of range -> zeros
            xl[it][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + q) * 16, 0, 0);
            xl[it][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + q + (Q >> 1)) * 16, 0, 0);
            xl[it][2] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (Q >> 1) - q) * 16, 0, 0);
            xl[it][3] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (Q >> 1) - q - 1) * 16, 0, 0);
            xl[it][4] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (q == 0 ? 0 : Q - q)) * 16, 0, 0);
            xl[it][5] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + Q - q - 1) * 16, 0, 0);
        }
    };
    auto store_chunk = [&](int c, int buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * NTHREADS + tid;
            const int ms = idx >> 4, ql = idx & 15;
            const f32x4 a = __builtin_bit_cast(f32x4, xl[it][0]), cc = __builtin_bit_cast(f32x4, xl[it][1]);
            const f32x4 b0 = __builtin_bit_cast(f32x4, xl[it][2]), b1 = __builtin_bit_cast(f32x4, xl[it][3]);
            const f32x4 d0 = __builtin_bit_cast(f32x4, xl[it][4]), d1 = __builtin_bit_cast(f32x4, xl[it][5]);
            const f32x4 b = f32x4{b0.x, b1.w, b1.z, b1.y}, d = f32x4{d0.x, d1.w, d1.z, d1.y};
            const f32x4 pe = a + cc, me = a - cc, qe = b + d, qo = b - d;
            f32x4 ue = pe + qe, ve = pe - qe, uo = me + qo, vo = me - qo;
            if (ql + 16 * c == 0) { ue.x = pe.x; ve.x = 0.f; uo.x = 0.f; vo.x = me.x; }     // j = 0 has no partner
            f32x4 *dst = lds + buf * BUF + ql * QS + ms;
            dst[0] = ue;
            dst[CH_ROWS * QS] = ve;
            dst[2 * CH_ROWS * QS] = uo;
            dst[3 * CH_ROWS * QS] = vo;
        }
    };

    // output rows 128 / 384 on the VALU: thread = (stream tid & 31, part tid >> 5); parts 0..3 dot ue with GSE[128], 4..7 uo
    // with GSO[128], four quads of every chunk each
    float r128 = 0.f;
    const bool do128 = NT == 2 || blockIdx.y == 0;
    const int part = tid >> 5, pr = part & 3, psel = part >> 2;

    // The operator stream does not depend on LDS, so its loads run D k-iterations (about 4 k cycles of MFMA) ahead,
    // across chunk boundaries.  That depth is what hides the next chunk's input loads: loads complete in order, so the
    // first operator block issued after them cannot be consumed before they have come back from HBM.
    constexpr int D = NT == 1 ? 8 : 4;
    f32x4 wq[D][NP], xq[2][NP];
    int ws = wbase + 8 + (NT == 2 ? 0 : 2 * po);           // 8 k-iterations x {SE, AE, SO, AO} per chunk
    const int xrow0 = (NT == 2 ? 0 : 2 * po) * CH_ROWS * QS + hq;
#define R_LDW(slot, j)                                                                          \
    _Pragma("unroll") for (int k = 0; k < NP; ++k) wq[slot][k] = ldw(wrs, lane16, ws + 4 * (j) + k);
#define R_LDX(slot, j)                                                                          \
    _Pragma("unroll") for (int k = 0; k < NP; ++k) xq[slot][k] = X[xrow0 + (k * CH_ROWS + 2 * (j)) * QS];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) { R_LDW(d, d) }
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const f32x4 *X = lds + (c & 1) * BUF;
        asm volatile("" : "+s"(ws));
        R_LDX(0, 0)
        f32x4 g128[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            R_LDW((j + D - 1) % D, j + D - 1)            // past j = 7: the next chunk's blocks (the stream is contiguous)
            if (j == 0) {
                if (c + 1 < nchunks) load_chunk(c + 1);  // global loads in flight under the MFMAs
                if (do128) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        g128[i] = ldw(wrs, (psel * (Q >> 2) + 16 * c + 4 * pr + i) * 16, (int)P.row128_block);
                }
            }
            if (j + 1 < 8) { R_LDX((j + 1) & 1, j + 1) }
            SB();
#pragma unroll
            for (int k = 0; k < NP; ++k) acc[k] = mfma4(wq[j % D][k], xq[j & 1][k], acc[k]);
            SB();
        }
        ws += 32;
        if (do128) {
            const int ms = tid & 31;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 uu = X[(psel * 2 * CH_ROWS + 4 * pr + i) * QS + ms];
                r128 += g128[i].x * uu.x + g128[i].y * uu.y + g128[i].z * uu.z + g128[i].w * uu.w;
            }
        }
        if (c + 1 < nchunks) store_chunk(c + 1, (c + 1) & 1);   // the other buffer: last read two chunks ago
        __syncthreads();
    }
#undef R_LDW
#undef R_LDX
    // epilogue: lane (m, h) holds rows o = 32 rt + 8g + 4h + i of its parts
    const int g2 = tile0 + m;
    float *const o = P.out + (size_t)g2 * 512;
    if constexpr (NT == 2) {
        if (g2 < P.n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 32 * rt + 8 * g + 4 * h;
                const f32x4 se = quad_of(acc[0], g), ae = quad_of(acc[1], g), so = quad_of(acc[2], g), ao = quad_of(acc[3], g);
                const f32x4 pe = se + ae, me = se - ae, pO = so + ao, mO = so - ao;
                *reinterpret_cast<f32x4 *>(o + row) = pe + pO;
                *reinterpret_cast<f32x4 *>(o + 256 + row) = pe - pO;
                const f32x4 lo = me + mO, hi = me - mO;
                if (row != 0) { o[256 - row] = lo.x; o[512 - row] = hi.x; }     // o = 0: rows 256 and 512 = 0 are written above
                o[255 - row] = lo.y; o[511 - row] = hi.y;
                o[254 - row] = lo.z; o[510 - row] = hi.z;
                o[253 - row] = lo.w; o[509 - row] = hi.w;
            }
        }
    } else {
        // partner waves (w ^ 1: the other pair of the same row tile) swap s + a and s - a through the now idle staging area
        f32x4 *const ex = lds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s_ = quad_of(acc[0], g), a_ = quad_of(acc[1], g);
            ex[(w * 8 + g) * 64 + lane] = s_ + a_;
            ex[(w * 8 + 4 + g) * 64 + lane] = s_ - a_;
        }
        __syncthreads();
        if (g2 < P.n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 32 * rt + 8 * g + 4 * h;
                const f32x4 s_ = quad_of(acc[0], g), a_ = quad_of(acc[1], g);
                const f32x4 pp = ex[((w ^ 1) * 8 + g) * 64 + lane], pm = ex[((w ^ 1) * 8 + 4 + g) * 64 + lane];
                if (po == 0) {          // this wave: (se, ae); partner: (so, ao) -> y[o], y[256 - o]
                    *reinterpret_cast<f32x4 *>(o + row) = (s_ + a_) + pp;
                    const f32x4 lo = (s_ - a_) + pm;
                    if (row != 0) o[256 - row] = lo.x;
                    o[255 - row] = lo.y; o[254 - row] = lo.z; o[253 - row] = lo.w;
                } else {                // this wave: (so, ao); partner: (se, ae) -> y[o + 256], y[512 - o]
                    *reinterpret_cast<f32x4 *>(o + 256 + row) = pp - (s_ + a_);
                    const f32x4 hi = pm - (s_ - a_);
                    if (row != 0) o[512 - row] = hi.x;
                    o[511 - row] = hi.y; o[510 - row] = hi.z; o[509 - row] = hi.w;
                }
            }
        }
    }
    if (do128) {
        const int ms = tid & 31;
        red[part * 32 + ms] = r128;
        __syncthreads();
        if (tid < 32 && tile0 + tid < P.n) {
            float e = 0.f, od = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { e += red[k * 32 + tid]; od += red[(4 + k) * 32 + tid]; }
            const float *x = P.in + (size_t)(tile0 + tid) * P.n_in;
            const f32x4 mid = ldw(wrs, (Q >> 1) * 16, (int)P.row128_block);       // floats 2Q, 2Q + 1: RE[128][Q] / 2, RO[128][Q] / 2
            e += mid.x * (x[Q] + x[3 * Q]);
            od += mid.y * (x[Q] - x[3 * Q]);
            P.out[(size_t)(tile0 + tid) * 512 + 128] = e + od;
            P.out[(size_t)(tile0 + tid) * 512 + 384] = e - od;
        }
    }
}

extern "C" hipError_t vadk_launch_resample(const vadk::ResampleParams *p, hipStream_t stream) {
    (void)hipGetLastError();   // HIP's last-error slot is sticky and process-wide: a stale failure from anywhere else must not become ours
    const int tiles = p->tile_start[p->nseg];
    if (tiles <= 0) return hipSuccess;
    // up to 256 chunk tiles: two workgroups per tile (finer grain also evens out mixed-rate launches, whose 48 kHz tiles run
    // six times longer than their 8 kHz ones)
    if (tiles <= 256)
        hipLaunchKernelGGL(vadk_resample_512<1>, dim3(tiles, 2), dim3(vadk::NTHREADS), 0, stream, *p);
    else
        hipLaunchKernelGGL(vadk_resample_512<2>, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p);
    return hipGetLastError();
}

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

// The operator is mirror-symmetric (R[o][i] == R[512-o][n_in-i]), so the kernel contracts the folded input
//   u[j] = x[j] + x[n_in-j],  v[j] = x[j] - x[n_in-j]   (j < n_in/2; u[0] = x[0], v[0] = 0)
// against the half-width operators GS / GA (pack_weights.cpp): Sh[o] = GS[o].u + R[o][n_in/2] x[n_in/2], Ah[o] = GA[o].v,
// y[o] = Sh + Ah, y[512-o] = Sh - Ah (o = 0..255), y[256] = Sh[256] - half the MFMAs of the dense product.
// NT = (S, A) tile pairs per wave.  NT = 2: one workgroup produces all 512 outputs of its 32 chunks (wave w: rows
// o = 64w..64w+63).  NT = 1: two workgroups (blockIdx.y) share a chunk tile (wave (y, w): o = 32 (4y + w)..+31) - used
// when the call has too few chunk tiles to fill the 256 CUs (the input is read twice, from L2).
template <int NT>
__global__ void __launch_bounds__(NTHREADS, 1) vadk_resample_512(const vadk::ResampleParams PP) {
    constexpr int CH_ROWS = 32;                       // one K chunk = 128 folded samples = 32 quad rows of u + 32 of v
    __shared__ f32x4 lds[2 * 2 * CH_ROWS * QS + 64];
    float *const red = reinterpret_cast<float *>(lds + 2 * 2 * CH_ROWS * QS);      // [8 parts][32 streams]: row 256
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
    const int K = P.n_in >> 1;                        // folded length
    const int nchunks = K >> 7;
    const int Q = P.n_in >> 2;                        // quads per input chunk
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.in), 0, (int)((unsigned)P.n * (unsigned)P.n_in * 4u), 0x00020000);
    const int lane16 = lane * 16;
    // this wave's rows: stream w' of the packed operator, first tile pair tp0 inside it
    const int ot0 = NT == 2 ? 2 * w : 4 * (int)blockIdx.y + w;       // first 32-row output tile (0..7)
    const int wbase = (ot0 >> 1) * (int)P.wave_blocks;
    const int tsel = ot0 & 1;                                         // NT = 1: which of the stream's two tiles

    // accumulators start from the rank-1 term of x[n_in/2] (the one sample the fold cannot pair): R[o][n_in/2] x[n_in/2]
    f32x16 accS[NT], accA[NT];
    {
        const int g2 = tile0 + m;
        const float xmid = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (g2 * P.n_in + K) * 4, 0, 0));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tb = wbase + 4 * (NT == 2 ? t : tsel);
            const f32x4 r0 = ldw(wrs, lane16, tb), r1 = ldw(wrs, lane16, tb + 1), r2 = ldw(wrs, lane16, tb + 2), r3 = ldw(wrs, lane16, tb + 3);
            accS[t] = f32x16{r0.x * xmid, r0.y * xmid, r0.z * xmid, r0.w * xmid, r1.x * xmid, r1.y * xmid, r1.z * xmid, r1.w * xmid,
                             r2.x * xmid, r2.y * xmid, r2.z * xmid, r2.w * xmid, r3.x * xmid, r3.y * xmid, r3.z * xmid, r3.w * xmid};
            accA[t] = (f32x16)(0.f);
        }
    }

    // chunk loader: 32 streams x 32 folded quads, 4 per thread; the three input quads of a folded quad are
    // x quad q (forward), quad Q - q (element 0) and quad Q - q - 1 (elements 3, 2, 1: x[n_in - 4q - e])
    u32x4 xa[4], xb[4], xc[4];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * NTHREADS + tid;
            const int ms = idx >> 5, q = (idx & 31) + 32 * c;
            const int base = (tile0 + ms) * Q;                       // streams past n: out of range -> zeros
            xa[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + q) * 16, 0, 0);
            xb[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (q == 0 ? 0 : Q - q)) * 16, 0, 0);
            xc[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + Q - q - 1) * 16, 0, 0);
        }
    };
    auto store_chunk = [&](int c, int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * NTHREADS + tid;
            const int ms = idx >> 5, ql = idx & 31;
            const f32x4 a = __builtin_bit_cast(f32x4, xa[it]), b = __builtin_bit_cast(f32x4, xb[it]), cc = __builtin_bit_cast(f32x4, xc[it]);
            f32x4 u = f32x4{a.x + b.x, a.y + cc.w, a.z + cc.z, a.w + cc.y};
            f32x4 v = f32x4{a.x - b.x, a.y - cc.w, a.z - cc.z, a.w - cc.y};
            if (ql + 32 * c == 0) { u.x = a.x; v.x = 0.f; }          // j = 0 has no partner
            lds[(buf * 2 * CH_ROWS + ql) * QS + ms] = u;
            lds[(buf * 2 * CH_ROWS + CH_ROWS + ql) * QS + ms] = v;
        }
    };

    // output row 256 on the VALU: thread = (stream tid & 31, part tid >> 5) sums 4 quads of every chunk
    float r256 = 0.f;
    const bool do256 = NT == 2 || blockIdx.y == 0;

    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const f32x4 *U = lds + (c & 1) * 2 * CH_ROWS * QS, *V = U + CH_ROWS * QS;
        if (c + 1 < nchunks) load_chunk(c + 1);          // global loads in flight under the MFMAs
        int ws = wbase + 8 + c * 64 + (NT == 2 ? 0 : tsel);    // 16 k-iterations x {S t0, S t1, A t0, A t1} per chunk
        asm volatile("" : "+s"(ws));
        if (do256) {
            const int ms = tid & 31, part = tid >> 5;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ql = part * 4 + i;
                const f32x4 g = ldw(wrs, (32 * c + ql) * 16, (int)P.row256_block);
                const f32x4 uu = U[ql * QS + ms];
                r256 += g.x * uu.x + g.y * uu.y + g.z * uu.z + g.w * uu.w;
            }
        }
        f32x4 AwS[NT], AwA[NT], BwS[NT], BwA[NT], Au, Av, Bu, Bv;
#define R_LD(S, j)                                                                              \
    _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                            \
        S##wS[k] = ldw(wrs, lane16, ws + 4 * (j) + k);                                          \
        S##wA[k] = ldw(wrs, lane16, ws + 4 * (j) + 2 + k);                                      \
    }                                                                                           \
    S##u = U[(2 * (j)) * QS + hq]; S##v = V[(2 * (j)) * QS + hq];
#define R_MMA(S)                                                                                \
    _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                            \
        accS[k] = mfma4(S##wS[k], S##u, accS[k]);                                               \
        accA[k] = mfma4(S##wA[k], S##v, accA[k]);                                               \
    }
        R_LD(A, 0)
        for (int j = 0; j < 16; j += 2) {
            R_LD(B, j + 1) SB();
            R_MMA(A) SB();
            const int jn = j + 2 < 16 ? j + 2 : 14;
            R_LD(A, jn) SB();
            R_MMA(B) SB();
        }
#undef R_LD
#undef R_MMA
        if (c + 1 < nchunks) store_chunk(c + 1, (c + 1) & 1);   // the other buffer: last read two chunks ago
        __syncthreads();
    }
    // epilogue: lane (m, h) holds rows o = 32 (ot0 + t) + 8g + 4h + i of Sh and Ah
    const int g2 = tile0 + m;
    if (g2 < P.n) {
        float *o = P.out + (size_t)g2 * 512;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 32 * (ot0 + t) + 8 * g + 4 * h;
                const f32x4 sh = quad_of(accS[t], g), ah = quad_of(accA[t], g);
                *reinterpret_cast<f32x4 *>(o + row) = f32x4{sh.x + ah.x, sh.y + ah.y, sh.z + ah.z, sh.w + ah.w};
                if (row != 0) o[512 - row] = sh.x - ah.x;            // y[512 - o] = Sh - Ah; o = 0 is its own mirror
                o[511 - row] = sh.y - ah.y;
                o[510 - row] = sh.z - ah.z;
                o[509 - row] = sh.w - ah.w;
            }
    }
    if (do256) {
        const int ms = tid & 31, part = tid >> 5;
        red[part * 32 + ms] = r256;
        __syncthreads();
        if (tid < 32 && tile0 + tid < P.n) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k * 32 + tid];
            const float xmid = P.in[(size_t)(tile0 + tid) * P.n_in + K];
            const float r = ldw(wrs, (K >> 2) * 16, (int)P.row256_block).x;     // K is a multiple of 128
            P.out[(size_t)(tile0 + tid) * 512 + 256] = s + r * xmid;
        }
    }
}

extern "C" hipError_t vadk_launch_resample(const vadk::ResampleParams *p, hipStream_t stream) {
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

// Batched Fourier resampler to 16 kHz for MI355X / gfx950.
//
// Replaces AudioUtils.resample_audio (/root/reference/src/real_time_vad/utils/audio.py:19-55 ->
// scipy.signal.resample, Fourier method, window=None) for chunks that yield exactly 512 output
// samples: 256 (8 kHz) / 768 (24 kHz) / 1536 (48 kHz) input samples per chunk.
//
// For a fixed (n_in, 512) the Fourier method is a fixed linear operator R[512][n_in]
// (SURVEY §8 a11); the host builds it in double precision from the closed form of scipy's
// spectrum copy (engine.cpp: build_resample_operator) and packs it like every other weight
// stream.  The kernel is the same MFMA skeleton as the model kernels: one workgroup = 32 streams,
// weights (R rows) on the A operand, the stream tile on the B operand, v_mfma_f32_32x32x2_f32.
// Wave w produces outputs 128w..128w+127 (4 tiles, 64 accumulator registers); the input is
// staged through LDS in chunks of 256 samples (64 quad rows), double buffered.
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

// NT = output tiles per wave.  NT = 4: one workgroup produces all 512 outputs of its 32 chunks.  NT = 2: two workgroups
// (blockIdx.y) share a chunk tile, 256 outputs each - used when the call has too few chunk tiles to fill the 256 CUs
// (the input is read twice, from L2, which costs less than idle CUs).
template <int NT>
__global__ void __launch_bounds__(NTHREADS, 1) vadk_resample_512(const vadk::ResampleParams PP) {
    constexpr int CH_ROWS = 64;                       // one chunk = 256 samples = 64 quad rows
    // which segment does this workgroup serve?  (block-uniform: scalar compares on kernel arguments)
    int sidx = 0;
#pragma unroll
    for (int k = 1; k < RESAMPLE_MAX_SEGS; ++k)
        if (k < PP.nseg && (int)blockIdx.x >= PP.tile_start[k]) sidx = k;
    const vadk::ResampleSeg P = PP.seg[sidx];
    const int tile_in_seg = (int)blockIdx.x - PP.tile_start[sidx];
    __shared__ f32x4 lds[2 * CH_ROWS * QS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int hq = h * QS + m;
    const int tile0 = tile_in_seg * MT;
    const int nchunks = P.n_in >> 8;
    const int quads_per_stream = P.n_in >> 2;
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wstream), 0, (int)P.wstream_bytes, 0x00020000);
    const int lane16 = lane * 16;
    // the packed operator: wave stream w' holds, per k-iteration, the blocks of output tiles 4w' .. 4w'+3.  With NT = 2 this
    // wave owns tiles 8y + 2w + {0, 1}: stream w' = 2y + (w >> 1), blocks 2(w & 1) + {0, 1} of each iteration.
    const int ot0 = NT == 4 ? 4 * w : 8 * (int)blockIdx.y + 2 * w;     // first output tile of this wave
    const int wbase = (ot0 >> 2) * (int)P.wave_blocks + (ot0 & 3);

    f32x16 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] = (f32x16)(0.f);

    // chunk loader: 32 streams x 64 quads = 2048 float4, 8 per thread; lanes run over quads of one stream
    f32x4 xr[8];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * NTHREADS + tid;
            const int ms = idx >> 6, q = idx & 63;
            const int g2 = tile0 + ms;
            xr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g2 < P.n) xr[it] = reinterpret_cast<const f32x4 *>(P.in)[(size_t)g2 * quads_per_stream + c * 64 + q];
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * NTHREADS + tid;
            lds[(buf * CH_ROWS + (idx & 63)) * QS + (idx >> 6)] = xr[it];
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const f32x4 *X = lds + (c & 1) * CH_ROWS * QS;
        if (c + 1 < nchunks) load_chunk(c + 1);          // global loads in flight under the MFMAs
        int ws = wbase + c * 128;                        // 32 k-iterations x 4 tiles per chunk
        asm volatile("" : "+s"(ws));
        f32x4 Aw[NT], Bw[NT], Aa, Ba;
#define R_LD(S, j)                                                                              \
    _Pragma("unroll") for (int k = 0; k < NT; ++k) S##w[k] = ldw(wrs, lane16, ws + 4 * (j) + k); \
    S##a = X[(2 * (j)) * QS + hq];
#define R_MMA(S) _Pragma("unroll") for (int k = 0; k < NT; ++k) acc[k] = mfma4(S##w[k], S##a, acc[k]);
        R_LD(A, 0)
        for (int j = 0; j < 32; j += 2) {
            R_LD(B, j + 1) SB();
            R_MMA(A) SB();
            const int jn = j + 2 < 32 ? j + 2 : 30;
            R_LD(A, jn) SB();
            R_MMA(B) SB();
        }
#undef R_LD
#undef R_MMA
        if (c + 1 < nchunks) store_chunk((c + 1) & 1);   // the other buffer: last read two chunks ago
        __syncthreads();
    }
    // epilogue: lane (m,h) holds outputs 32*(ot0+k) + 8g + 4h + i
    const int g2 = tile0 + m;
    if (g2 < P.n) {
        float *o = P.out + (size_t)g2 * 512 + 32 * ot0 + 4 * h;
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(o + 32 * k + 8 * g) = quad_of(acc[k], g);
    }
}

extern "C" hipError_t vadk_launch_resample(const vadk::ResampleParams *p, hipStream_t stream) {
    const int tiles = p->tile_start[p->nseg];
    if (tiles <= 0) return hipSuccess;
    // up to 256 chunk tiles: two workgroups per tile (finer grain also evens out mixed-rate launches, whose 48 kHz tiles run
    // six times longer than their 8 kHz ones)
    if (tiles <= 256)
        hipLaunchKernelGGL(vadk_resample_512<2>, dim3(tiles, 2), dim3(vadk::NTHREADS), 0, stream, *p);
    else
        hipLaunchKernelGGL(vadk_resample_512<4>, dim3(tiles), dim3(vadk::NTHREADS), 0, stream, *p);
    return hipGetLastError();
}

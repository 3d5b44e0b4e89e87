// Micro-benchmark for DESIGN.md 7.5: what MFMA rate does a weight-streaming wave sustain with 32-stream tiles
// (v_mfma_f32_32x32x2_f32, one 256-thread workgroup per CU) against 16-stream tiles (v_mfma_f32_16x16x4_f32, two
// workgroups per CU = two waves per SIMD, every 1 KiB operand block serving half as many streams)?
// Each wave streams BLOCKS operand blocks of 1 KiB from a 1.2 MB buffer (L2-resident, like the packed weights), 4 MFMAs per
// block on NACC rotating accumulators, plus NV dependent VALU ops per block (the "epilogue" work that cannot overlap the
// same wave's MFMAs).  MODE 32: grid = 256 x 256 threads, LDS 80 KB+ so ONE workgroup per CU.  MODE 16: grid = 512, LDS
// 70 KB so TWO per CU.  Prints cycles per block per wave and the fraction of the fp32 MFMA peak the chip sustains.
// Developer tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef MODE
#define MODE 32
#endif
#ifndef NV
#define NV 8
#endif
constexpr int LDS_QUADS = MODE == 32 ? 6000 : 4400;     // 96 KB -> one workgroup per CU; 70 KB -> two
__global__ void __launch_bounds__(256, 1) k(float *out, const float *w, unsigned wbytes, int blocks, unsigned long long *ticks) {
    __shared__ f32x4 lds[LDS_QUADS];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, (int)wbytes, 0x00020000);
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const f32x4 b = lds[threadIdx.x];
    float z = b.x;
    const int nblk = (int)(wbytes / 1024);
    int blk = (blockIdx.x * 4 + wv) * 37 % nblk;
    unsigned long long t0 = clock64();
    // operand blocks are requested D ahead (a ring of registers, the loop unrolled over it), as the kernels do
    constexpr int D = 8;
    f32x4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, blk * 1024, 0));
        blk = blk + 1 < nblk ? blk + 1 : 0;
    }
#if MODE == 32
    f32x16 acc[2] = {(f32x16)(0.f), (f32x16)(0.f)};
    f32x16 accx[4] = {(f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f)};
    for (int i = 0; i < blocks; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const f32x4 cur = ring[d];
            ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, blk * 1024, 0));
            blk = blk + 1 < nblk ? blk + 1 : 0;
#ifdef INDEP      // the four MFMAs of a block on four accumulators (no dependent pair back to back)
            accx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x, b.x, accx[0], 0, 0, 0);
            accx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.y, b.y, accx[1], 0, 0, 0);
            accx[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.z, b.z, accx[2], 0, 0, 0);
            accx[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.w, b.w, accx[3], 0, 0, 0);
#else
            f32x16 a = acc[d & 1];
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x, b.x, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.y, b.y, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.z, b.z, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.w, b.w, a, 0, 0, 0);
            acc[d & 1] = a;
#endif
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z) : "v"(b.y));
        }
    }
    float s = z;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r] + accx[0][r] + accx[1][r] + accx[2][r] + accx[3][r];
#else
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int i = 0; i < blocks; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const f32x4 cur = ring[d];
            ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, blk * 1024, 0));
            blk = blk + 1 < nblk ? blk + 1 : 0;
            f32x4 a = acc[d & 3];
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x, b.x, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y, b.y, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.z, b.z, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.w, b.w, a, 0, 0, 0);
            acc[d & 3] = a;
#pragma unroll
            for (int v = 0; v < NV / 2; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z) : "v"(b.y));   // half the streams: half the epilogue work
        }
    }
    float s = z;
    for (int r = 0; r < 4; ++r) s += acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r];
#endif
    unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * 4 + wv] = t1 - t0;
}
int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1200;
    const int grid = MODE == 32 ? 256 : 512;
    const unsigned wbytes = 1234 * 1024;
    float *w, *out; unsigned long long *tk;
    hipMalloc(&w, wbytes); hipMalloc(&out, grid * 256 * 4); hipMalloc(&tk, grid * 4 * 8);
    std::vector<float> h(wbytes / 4); for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, w, wbytes, blocks, tk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(grid * 4); hipMemcpy(t.data(), tk, t.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : t) avg += (double)v; avg /= t.size();
        const double flop = (double)grid * 4 * blocks * 4 * (MODE == 32 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4);
        printf("MODE %d NV %d: %.1f us, %.0f cycles per block per wave (MFMA alone: %d), %.1f TFLOP/s = %.3f of 157.3, operand traffic %.1f TB/s\n",
               MODE, NV, ms * 1e3, avg / blocks, MODE == 32 ? 256 : 128, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 157.3e12,
               (double)grid * 4 * blocks * 1024 / (ms * 1e-3) / 1e12);
    }
    return 0;
}

// Micro-benchmark (developer tool, not product): what would the split-operand form of DESIGN.md §7 (3b) buy?
// A contraction in the shape of the model kernels' inner loops on 16-stream tiles - per wave two 16-row output tiles, weights
// streamed from L2 in 1 KiB blocks through one buffer descriptor (A operand), activations read from LDS (B operand), software
// pipelined one step ahead - in two forms over the same K:
//   fp32  : v_mfma_f32_16x16x4_f32, per K = 32: 4 weight blocks, 2 LDS reads, 16 MFMAs
//   bf16x6: operands as three bf16 pieces each (weights pre-split, activations pre-split in LDS), the six leading cross terms on
//           v_mfma_f32_16x16x32_bf16: per K = 32: 6 weight blocks, 3 LDS reads, 12 MFMAs
// and the VALU cost of splitting a float into its three pieces.  Prints cycles per K = 32 step per wave and the ratio.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ub tools/ubench_split_bf16.cpp && /tmp/ub [workgroups_per_cu=1]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int STEPS = 256;            // K = 32 steps per wave and launch (a model kernel has ~150)
constexpr int WBLOCKS = 1200;         // weight footprint cycled through: 1.2 MB, L2-resident like the model's streams

__device__ __forceinline__ u32x4 ldw(__amdgpu_buffer_rsrc_t rs, int voff, int blk) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, blk * 1024, 0);
}

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_fp32(const float *W, float *out, unsigned long long *cyc) {
    __shared__ f32x4 lds[64 * 16 * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 64 * 16 * 4; i += 256) lds[i] = f32x4{0.001f * i, 0.5f, -0.25f, 1.f};
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, WBLOCKS * 1024, 0x00020000);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};     // four independent chains, as the kernels have (>= 4 accumulators)
    int blk = (blockIdx.x * 4 + w) * 37 % (WBLOCKS - 8);
    u32x4 Aw[4], Bw[4];
    f32x4 Aa[2], Ba[2];
#define LD(S, st)                                                                                           \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) S##w[k] = ldw(rs, lane * 16, blk + k);                    \
    blk = blk + 4 >= WBLOCKS - 4 ? 0 : blk + 4;                                                             \
    S##a[0] = lds[(((st) * 8) & 255) * 16 + lane]; S##a[1] = lds[(((st) * 8 + 4) & 255) * 16 + lane];
#define MMA(S)                                                                                              \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                          \
        const f32x4 w0 = __builtin_bit_cast(f32x4, S##w[2 * h]), w1 = __builtin_bit_cast(f32x4, S##w[2 * h + 1]);   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
            acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i], S##a[h][i], acc[i & 1], 0, 0, 0);       \
            acc[2 + (i & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[i], S##a[h][i], acc[2 + (i & 1)], 0, 0, 0); \
        }                                                                                                   \
    }
    const unsigned long long t0 = clock64();
    LD(A, 0)
    for (int st = 0; st < STEPS; st += 2) {
        LD(B, st + 1) SB();
        MMA(A) SB();
        LD(A, st + 2) SB();
        MMA(B) SB();
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
#undef LD
#undef MMA
}

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_bf16x6(const float *W, float *out, unsigned long long *cyc) {
    __shared__ f32x4 lds[64 * 16 * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 64 * 16 * 4; i += 256) lds[i] = f32x4{0.001f * i, 0.5f, -0.25f, 1.f};
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, WBLOCKS * 1024, 0x00020000);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int blk = (blockIdx.x * 4 + w) * 37 % (WBLOCKS - 8);
    u32x4 Aw[6], Bw[6];          // row tile rt, piece p -> [3 rt + p]
    u32x4 Aa[3], Ba[3];          // activation pieces
#define LD(S, st)                                                                                           \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) S##w[k] = ldw(rs, lane * 16, blk + k);                    \
    blk = blk + 6 >= WBLOCKS - 6 ? 0 : blk + 6;                                                             \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) S##a[p] = __builtin_bit_cast(u32x4, lds[(((st) * 12 + 4 * p) & 255) * 16 + lane]);
#define X(wv, av, a_) a_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, av), a_, 0, 0, 0);
    // six leading cross terms: (w1 x1) (w1 x2) (w2 x1) (w1 x3) (w2 x2) (w3 x1)
#define MMA(S)                                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                                       \
        X(S##w[3 * rt], S##a[0], acc[2 * rt]) X(S##w[3 * rt], S##a[1], acc[2 * rt + 1]) X(S##w[3 * rt + 1], S##a[0], acc[2 * rt])   \
        X(S##w[3 * rt], S##a[2], acc[2 * rt + 1]) X(S##w[3 * rt + 1], S##a[1], acc[2 * rt]) X(S##w[3 * rt + 2], S##a[0], acc[2 * rt + 1]) \
    }
    const unsigned long long t0 = clock64();
    LD(A, 0)
    for (int st = 0; st < STEPS; st += 2) {
        LD(B, st + 1) SB();
        MMA(A) SB();
        LD(A, st + 2) SB();
        MMA(B) SB();
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
#undef LD
#undef MMA
#undef X
}

// VALU cost of the split: x -> (bf16 x1, bf16 x2, bf16 x3), 64 values per lane
__global__ void __launch_bounds__(256) k_split(const float *in, unsigned *out, unsigned long long *cyc) {
    const int tid = threadIdx.x;
    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = in[(blockIdx.x * 64 + i) * 256 + tid];
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(v[i]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t0 = clock64();
    asm volatile("" : "+v"(v[0]), "+s"(const_cast<unsigned long long &>(t0)));
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const __bf16 a1 = (__bf16)v[i];
        const float r1 = v[i] - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;
        const __bf16 a3 = (__bf16)r2;
        o ^= (unsigned)__builtin_bit_cast(unsigned short, a1) ^ ((unsigned)__builtin_bit_cast(unsigned short, a2) << 8) ^
             ((unsigned)__builtin_bit_cast(unsigned short, a3) << 16);
    }
    asm volatile("" : "+v"(o));
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = o;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

int main(int argc, char **argv) {
    const int occ = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = 256 * occ;
    float *W, *out, *in;
    unsigned *o2;
    unsigned long long *cyc;
    CK(hipMalloc(&W, WBLOCKS * 1024));
    CK(hipMemset(W, 0, WBLOCKS * 1024));
    CK(hipMalloc(&out, blocks * 256 * 4));
    CK(hipMalloc(&o2, blocks * 256 * 4));
    CK(hipMalloc(&in, (size_t)blocks * 64 * 256 * 4));
    CK(hipMemset(in, 0, (size_t)blocks * 64 * 256 * 4));
    CK(hipMalloc(&cyc, blocks * 4 * 8));
    std::vector<unsigned long long> h(blocks * 4);
    auto report = [&](const char *name, double per) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%-28s %8.1f cycles per %s (mean over %zu waves)\n", name, s / h.size() / per, per == STEPS ? "K = 32 step" : "value", h.size());
        return s / h.size() / per;
    };
    double f = 0, b = 0;
    for (int rep = 0; rep < 3; ++rep) {
        if (occ == 1) hipLaunchKernelGGL(k_fp32<1>, dim3(blocks), dim3(256), 0, 0, W, out, cyc); else hipLaunchKernelGGL(k_fp32<2>, dim3(blocks), dim3(256), 0, 0, W, out, cyc);
        f = report("fp32 16x16x4, 16 MFMAs/step", STEPS);
        if (occ == 1) hipLaunchKernelGGL(k_bf16x6<1>, dim3(blocks), dim3(256), 0, 0, W, out, cyc); else hipLaunchKernelGGL(k_bf16x6<2>, dim3(blocks), dim3(256), 0, 0, W, out, cyc);
        b = report("bf16x6 16x16x32, 12 MFMAs/step", STEPS);
    }
    printf("workgroups per CU: %d   fp32 / bf16x6 = %.2f\n", occ, f / b);
    hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, in, o2, cyc);
    report("split into three bf16 pieces", 64);
    return 0;
}

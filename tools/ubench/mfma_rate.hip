// Micro-benchmark: sustained issue rate of v_mfma_f32_32x32x2_f32 with NACC independent accumulators,
// one wave per SIMD (256 threads per WG, 1 WG per CU).  Developer tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef NACC
#define NACC 6
#endif
__global__ void __launch_bounds__(256, 1) k(float *out, const float *in, int iters, unsigned long long *ticks) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = (f32x16)(0.f);
    float x[4], y[4];
    for (int i = 0; i < 4; ++i) { x[i] = in[threadIdx.x * 4 + i]; y[i] = in[1024 + threadIdx.x * 4 + i]; }
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], y[(i + a) & 3], acc[a], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 256, iters = 2000;
    float *in, *out; unsigned long long *tk;
    hipMalloc(&in, 2048 * 4); hipMalloc(&out, grid * 256 * 4); hipMalloc(&tk, grid * 8);
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, 256>>>(out, in, iters, tk); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<grid, 256>>>(out, in, iters, tk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 4 * NACC;
    printf("NACC=%d grid=%d: %.1f ticks/MFMA, %.2f us, %.1f TFLOP/s, tick rate %.2f GHz\n", NACC, grid, t / n, ms * 1e3,
           n * 4096.0 * 4 * grid / (ms * 1e-3) / 1e12, t / (ms * 1e-3) / 1e9);
    return 0;
}

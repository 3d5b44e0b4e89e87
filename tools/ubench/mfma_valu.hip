// Micro-benchmark: how much independent work of one kind fits in the shadow of a v_mfma_f32_32x32x2_f32 issued by
// the SAME wave (one wave per SIMD).  KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_accvgpr_read_b32, 3 ds_write_b128,
// 4 v_exp_f32, 5 v_cndmask_b32 (VCC read), 6 ds_read_b128.  NV = instructions of that kind after every MFMA.
// Developer tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef KIND
#define KIND 0
#endif
#ifndef NV
#define NV 4
#endif
__global__ void __launch_bounds__(256, 1) k(float *out, const float *in, int iters, unsigned long long *ticks) {
    __shared__ f32x4 lds[256 * 4];
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) acc[a] = (f32x16)(0.f);
    float x = in[threadIdx.x], y = in[256 + threadIdx.x];
    float z[16];
    f32x2 p[16];
    f32x4 d = {x, y, x, y};
    for (int i = 0; i < 16; ++i) { z[i] = in[512 + i]; p[i] = f32x2{z[i], x}; }
    f32x2 px = {x, y};
    f32x4 *my = lds + threadIdx.x;
    lds[threadIdx.x] = d;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[a]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int j = (a * NV + i) & 15;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[j]) : "v"(x), "v"(y));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(px), "v"(px));
                if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(z[j]) : "a"(acc[(a + 2) & 3][i & 15]));
                if (KIND == 3) asm volatile("ds_write_b128 %0, %1" :: "v"((int)(threadIdx.x * 16 + (i & 3) * 4096)), "v"(d) : "memory");
                if (KIND == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(z[j]) : "v"(x));
                if (KIND == 5) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(z[j]) : "v"(x), "v"(y));
                if (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"((int)(threadIdx.x * 16 + (i & 3) * 4096)) : "memory");
            }
        }
        if (KIND == 3 || KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = clock64();
    float s = d.x + d.y;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += z[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + my[0].x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv) {
    int grid = 256, iters = 1000;
    float *in, *out; unsigned long long *tk;
    hipMalloc(&in, 2048 * 4); hipMalloc(&out, grid * 256 * 4); hipMalloc(&tk, grid * 8);
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    k<<<grid, 256>>>(out, in, iters, tk); hipDeviceSynchronize();
    k<<<grid, 256>>>(out, in, iters, tk); hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    printf("KIND=%d NV=%d: %.1f ticks per (MFMA + %d)\n", KIND, NV, t / ((double)iters * 4), NV);
    return 0;
}

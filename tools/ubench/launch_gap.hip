// Micro-benchmark: cost of one back-to-back launch in a stream for a do-nothing kernel with the V5 kernel's
// shape (256 workgroups x 256 threads, 121 KB LDS, ~350-byte kernarg).  Developer tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { void *p[8]; int sect[4][16]; int n, T; };
__global__ void __launch_bounds__(256, 1) nop_kernel(const Args a, int *out) {
    extern __shared__ float lds[];
    if (a.n == -1) { lds[threadIdx.x] = 1.f; out[threadIdx.x] = (int)lds[(threadIdx.x + 1) & 255]; }
}
int main() {
    Args a{}; a.n = 8192; a.T = 1;
    int *out; hipMalloc(&out, 4096);
    hipStream_t s; hipStreamCreate(&s);
    hipFuncSetAttribute((const void *)nop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 121632);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {0, 121632}) {
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 1000; ++i) nop_kernel<<<256, 256, lds, s>>>(a, out);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("lds=%d: %.2f us per empty launch\n", lds, ms);
        }
    }
    return 0;
}

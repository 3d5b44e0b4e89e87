// Does hipExtAnyOrderLaunch let a kernel start before the previous kernel of the same stream has finished on gfx950?
// (hip_ext.h says "not supported on AMD GFX9xx boards" for the module-launch form.)  Kernel A spins ~200 us on one workgroup and
// stamps its end; kernel B (launched with the flag right behind it) stamps its start.  B_start < A_end <=> the barrier bit is gone.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long *t, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) t[0] = wall_clock64();
}
__global__ void stamp(unsigned long long *t) { if (threadIdx.x == 0) t[1] = wall_clock64(); }
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int flags = 0; flags < 2; ++flags) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 20000LL);     // wall_clock64: 100 MHz -> 200 us
        hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, d);
        hipStreamSynchronize(s);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("flags=%d  A_end=%llu  B_start=%llu  B_start - A_end = %lld ticks (negative: B overlapped A)\n", flags, h[0], h[1], (long long)(h[1] - h[0]));
    }
    return 0;
}

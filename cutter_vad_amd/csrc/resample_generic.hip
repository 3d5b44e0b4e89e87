// Generic whole-array Fourier resampler (resample_generic.h): y = R x for ANY (n_in -> n_out), R never stored.
//
// AudioUtils.resample_audio (/root/reference/src/real_time_vad/utils/audio.py:39-49) on arrays that are not one of the three
// streaming chunk shapes (those have their own MFMA kernel, resample.hip).  One array is a matrix-VECTOR product: there is
// nothing for the matrix cores to reuse, and a stored operator would be n_in * n_out * 4 bytes of HBM traffic per call
// (3 GB for one second of 48 kHz audio).  So every entry is evaluated where it is used, in float64, from tables that are
// a few bytes per sample:
//
//   * a lane owns ONE output m (its five table values stay in registers), a wave walks a slice of n;
//   * x[n] and the four table values of n are wave-uniform: scalar loads, 36 bytes per step for 64 outputs;
//   * per entry: two float64 products + two FMAs (both sines of a difference), one division, one FMA into the float64 sum;
//     (m a - n b) mod L is carried along in integers - it finds t == 0 exactly, and the small-angle series of sin(pi t)
//     takes over where the difference of products would lose its leading digits;
//   * slices of one output are summed in a fixed order by the finishing kernel: results do not depend on the launch shape.
//
// Bound: float64 VALU (about 25 issue slots per entry), not HBM - the tables of a call fit in L2.
#include <hip/hip_runtime.h>

#include "resample_generic.h"

namespace {

using vadk::RsgParams;

template <bool XF64>
__global__ __launch_bounds__(256) void vadk_rsg_partial(const RsgParams p) {
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.y * 4 + (threadIdx.x >> 6);          // wave-uniform
    if (slice >= p.nslice) return;
    const int row = blockIdx.z;
    const int64_t m = p.m_begin + (int64_t)blockIdx.x * 64 + lane;
    const bool live = m < p.m_end;
    const int64_t mm = live ? m : p.m_end - 1;
    const int64_t n_lo = (int64_t)slice * p.slice_len;
    const int64_t n_hi = n_lo + p.slice_len < p.n_in ? n_lo + p.slice_len : p.n_in;

    const double *__restrict__ tm = p.tm + mm * 5;
    const double sA = tm[0], cA = tm[1], sa = tm[2], ca = tm[3], Cm = p.corrected ? tm[4] : 0.0;
    const int64_t L = p.L, b = p.b, halfL = L >> 1;
    int64_t j = mm * p.a - n_lo * b;                                 // both products are below L: j in (-L, L)
    if (j < 0) j += L;
    const double invL = 1.0 / (double)L;
    const double *__restrict__ tn = p.tn;
    const float *__restrict__ xf = static_cast<const float *>(p.x) + (size_t)row * p.n_in;
    const double *__restrict__ xd = static_cast<const double *>(p.x) + (size_t)row * p.n_in;

    double acc = 0.0;
    for (int64_t n = n_lo; n < n_hi; ++n) {
        const double cB = tn[n * 4 + 0], sB = tn[n * 4 + 1], cb = tn[n * 4 + 2], sb = tn[n * 4 + 3];
        const double xv = XF64 ? xd[n] : (double)xf[n];
        const double num = __builtin_fma(sA, cB, -(cA * sB));        // sin(A_m - B_n)
        double den = __builtin_fma(sa, cb, -(ca * sb));              // sin(a_m - b_n)
        const int64_t jc = j > halfL ? j - L : j;
        const double r = (double)jc * invL;
        if (__builtin_fabs(r) < 0x1p-10) {
            const double w = 3.14159265358979323846264338327950288 * r, u = w * w;
            const double ser = w * (1.0 - u * (1.0 / 6.0) * (1.0 - u * (1.0 / 20.0) * (1.0 - u * (1.0 / 42.0))));
            den = __builtin_copysign(__builtin_fabs(ser), den);
        }
        double d = j == 0 ? p.peak : num / den;
        if (j != 0) d -= (n & 1) ? -Cm : Cm;
        acc = __builtin_fma(xv, d, acc);
        j -= b;
        if (j < 0) j += L;
    }
    if (live) p.partial[((size_t)slice * p.rows + row) * (size_t)(p.m_end - p.m_begin) + (size_t)(m - p.m_begin)] = acc;
}

__global__ __launch_bounds__(256) void vadk_rsg_finish(const RsgParams p) {
    const int64_t w = p.m_end - p.m_begin;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;
    if (i >= w) return;
    double s = 0.0;
    for (int k = 0; k < p.nslice; ++k) s += p.partial[((size_t)k * p.rows + row) * (size_t)w + (size_t)i];
    p.y[(size_t)row * p.n_out + (size_t)(p.m_begin + i)] = (float)(s * p.inv_n_in);
}

}  // namespace

extern "C" hipError_t vadk_launch_rsg_partial(const vadk::RsgParams *p, hipStream_t stream) {
    const int64_t w = p->m_end - p->m_begin;
    if (w < 1 || p->rows < 1 || p->rows > 65535 || p->nslice < 1) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((w + 63) / 64), (unsigned)((p->nslice + 3) / 4), (unsigned)p->rows);
    if (grid.y > 65535) return hipErrorInvalidValue;
    if (p->x_f64) hipLaunchKernelGGL(vadk_rsg_partial<true>, grid, dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(vadk_rsg_partial<false>, grid, dim3(256), 0, stream, *p);
    return hipGetLastError();
}

extern "C" hipError_t vadk_launch_rsg_finish(const vadk::RsgParams *p, hipStream_t stream) {
    const int64_t w = p->m_end - p->m_begin;
    if (w < 1 || p->rows < 1 || p->rows > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vadk_rsg_finish, dim3((unsigned)((w + 255) / 256), (unsigned)p->rows), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

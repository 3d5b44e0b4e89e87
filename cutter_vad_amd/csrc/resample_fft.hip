// Whole-array Fourier resampling for LONG arrays: the same function as resample_generic.hip (AudioUtils.resample_audio,
// /root/reference/src/real_time_vad/utils/audio.py:39-49 -> scipy.signal.resample(x, n_out), real input, no window), as two chirp-z
// (Bluestein) transforms on power-of-two float64 FFTs, so that the work is O(n log n) for ANY pair of lengths.
//
//   forward:  X[k] = sum_n x[n] e^{-2 pi i k n / N},  k = 0 .. K = min(N, M) // 2        (the bins scipy keeps)
//             k n = (k^2 + n^2 - (k - n)^2) / 2   =>   X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k - n],   c[n] = e^{-i pi n^2 / N}
//   edit:     Z[k] = g_k Y[k]: Y = X with scipy's Nyquist rule (bin N'/2 of N' = min(N, M) even: x 2 when shrinking, x 1/2 when
//             growing), g_0 = 1, g_k = 2, g_{M/2} = 1 (irfft reads the half spectrum: every bin but DC and M/2 counts twice)
//   inverse:  y[m] = 1/N Re sum_{k <= K} Z[k] e^{+2 pi i k m / M}  =  1/N Re( d[m] * sum_k (Z[k] d[k]) conj(d)[m - k] ),  d[n] = e^{+i pi n^2 / M}
// Each sum is a linear convolution = FFT_P, pointwise product with the (cached) transform of the chirp kernel, inverse FFT_P, for
// a power of two P >= the convolution's span.  Chirp phases are reduced in integers (n^2 mod 2N in 64 bits) before sincospi, and
// everything is float64 until the final cast, so the result is scipy's to float32 rounding (tests: <= 3e-7, as the direct kernel).
//
// FFT: Stockham autosort, radix 8 in registers (a radix-4 / radix-2 stage first for the remainder of log2 P), one pass over HBM per
// stage (reads contiguous, writes in runs of the stage's stride; the first stage writes adjacent elements per thread).  HBM-bound:
// 2 x 16 P bytes per stage, ceil(log2(P) / 3) stages; a 60 s / 48 kHz array (P = 2^23) moves ~2.1 GB per transform.  Measured with
// radix-2 stages only: 5.0 TB/s on a ten-minute array, i.e. the passes run near copy speed and the lever is the NUMBER of passes:
// 36.7 ms (radix 2) -> 23 ms (radix 4) -> 18.5 ms (radix 8) for that array.  The arrays this path exists for are off the serving
// path, so plain passes were preferred to LDS-blocked ones.
#include <hip/hip_runtime.h>

#include "resample_generic.h"

namespace {

using vadk::RsfParams;
struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// e^{sign i pi n^2 / len}: n^2 mod 2 len in integers (n < 2^31)
__device__ __forceinline__ cplx chirp(int64_t n, int64_t len, double sign) {
    const uint64_t r = ((uint64_t)n * (uint64_t)n) % (uint64_t)(2 * len);
    double s, c;
    sincospi((double)r / (double)len, &s, &c);
    return {c, sign * s};
}

// W[j] = e^{-2 pi i j / P}, j < P / 2
__global__ __launch_bounds__(256) void vadk_rsf_twiddles(cplx *W, int64_t P) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= P / 2) return;
    double s, c;
    sincospi(2.0 * (double)j / (double)P, &s, &c);
    W[j] = {c, -s};
}

// the chirp kernel of a transform, wrapped into P entries: b[n] = b[P - n] = conj(e^{sign i pi n^2 / len}) for n < span, else 0
__global__ __launch_bounds__(256) void vadk_rsf_kernel(cplx *b, int64_t P, int64_t len, int64_t span, double sign) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int64_t n = i < span ? i : (P - i < span ? P - i : -1);
    cplx v = {0.0, 0.0};
    if (n >= 0) {
        v = chirp(n, len, sign);
        v.im = -v.im;
    }
    b[i] = v;
}

// a[n] = x[n] c[n] (n < N), 0 up to P; one row per blockIdx.y
template <bool XF64>
__global__ __launch_bounds__(256) void vadk_rsf_load(const RsfParams p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.P1) return;
    const int row = blockIdx.y;
    cplx v = {0.0, 0.0};
    if (i < p.n_in) {
        const double x = XF64 ? static_cast<const double *>(p.x)[(size_t)row * p.n_in + i] : (double)static_cast<const float *>(p.x)[(size_t)row * p.n_in + i];
        const cplx c = chirp(i, p.n_in, -1.0);
        v = {x * c.re, x * c.im};
    }
    static_cast<cplx *>(p.a)[(size_t)row * p.Pmax + i] = v;
}

// one Stockham radix-2 stage: n = current sub-transform length, s = stride (n * s = P); src -> dst
__global__ __launch_bounds__(256) void vadk_rsf_stage(const cplx *__restrict__ src, cplx *__restrict__ dst, const cplx *__restrict__ W,
                                                      int64_t P, int64_t s, int64_t Pmax) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= P / 2) return;
    const size_t row = (size_t)blockIdx.y * (size_t)Pmax;
    const int64_t pp = t / s, q = t - pp * s;
    const cplx a = src[row + t], b = src[row + t + P / 2];
    const cplx w = W[pp * s];                       // e^{-2 pi i pp / n} = W_P[pp * (P / n)] = W_P[pp * s]
    const cplx d = {a.re - b.re, a.im - b.im};
    dst[row + q + 2 * s * pp] = {a.re + b.re, a.im + b.im};
    dst[row + q + 2 * s * pp + s] = cmul(d, w);
}

// one Stockham radix-4 stage (n = P / s, n1 = n / 4): a quarter of the threads of a radix-2 stage move twice the elements each, and
// the array crosses HBM half as often.  a, b, c, d = the four quarter-strided inputs; w1 = e^{-2 pi i p / n}, w2 = w1^2, w3 = w1 w2
__global__ __launch_bounds__(256) void vadk_rsf_stage4(const cplx *__restrict__ src, cplx *__restrict__ dst, const cplx *__restrict__ W,
                                                       int64_t P, int64_t s, int64_t Pmax) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t Q = P / 4;
    if (t >= Q) return;
    const size_t row = (size_t)blockIdx.y * (size_t)Pmax;
    const int64_t pp = t / s, q = t - pp * s;
    const cplx a = src[row + t], b = src[row + t + Q], c = src[row + t + 2 * Q], d = src[row + t + 3 * Q];
    const cplx w1 = W[pp * s], w2 = W[2 * pp * s], w3 = cmul(w1, w2);
    const cplx apc = {a.re + c.re, a.im + c.im}, amc = {a.re - c.re, a.im - c.im};
    const cplx bpd = {b.re + d.re, b.im + d.im}, jbmd = {-(b.im - d.im), b.re - d.re};       // i (b - d)
    cplx *o = dst + row + q + 4 * s * pp;
    o[0] = {apc.re + bpd.re, apc.im + bpd.im};
    o[s] = cmul(w1, cplx{amc.re - jbmd.re, amc.im - jbmd.im});
    o[2 * s] = cmul(w2, cplx{apc.re - bpd.re, apc.im - bpd.im});
    o[3 * s] = cmul(w3, cplx{amc.re + jbmd.re, amc.im + jbmd.im});
}

// one Stockham radix-8 stage (n = P / s, n1 = n / 8): eight eighth-strided inputs per thread, an 8-point DFT in registers (three
// radix-2 levels, decimation in frequency), output k scaled by w^k, w = e^{-2 pi i p / n}: a third of the passes of radix 2
__global__ __launch_bounds__(256) void vadk_rsf_stage8(const cplx *__restrict__ src, cplx *__restrict__ dst, const cplx *__restrict__ W,
                                                       int64_t P, int64_t s, int64_t Pmax) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t E = P / 8;
    if (t >= E) return;
    const size_t row = (size_t)blockIdx.y * (size_t)Pmax;
    const int64_t pp = t / s, q = t - pp * s;
    cplx a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = src[row + t + j * E];
    auto add = [](cplx x, cplx y) { return cplx{x.re + y.re, x.im + y.im}; };
    auto sub = [](cplx x, cplx y) { return cplx{x.re - y.re, x.im - y.im}; };
    auto mulmi = [](cplx x) { return cplx{x.im, -x.re}; };                 // x * (-i)
    const double h = 0.70710678118654752440;
    const cplx b0 = add(a[0], a[4]), b4 = sub(a[0], a[4]), b1 = add(a[1], a[5]), b5 = sub(a[1], a[5]);
    const cplx b2 = add(a[2], a[6]), b6 = sub(a[2], a[6]), b3 = add(a[3], a[7]), b7 = sub(a[3], a[7]);
    // even outputs: the 4-point DFT of b0 .. b3
    const cplx c0 = add(b0, b2), c2 = sub(b0, b2), c1 = add(b1, b3), c3 = mulmi(sub(b1, b3));
    // odd outputs: the 4-point DFT of b4, b5 w8, b6 w8^2, b7 w8^3   (w8 = (1 - i) / sqrt 2)
    const cplx d1 = {h * (b5.re + b5.im), h * (b5.im - b5.re)}, d2 = mulmi(b6), d3 = {h * (b7.im - b7.re), -h * (b7.re + b7.im)};
    const cplx e0 = add(b4, d2), e2 = sub(b4, d2), e1 = add(d1, d3), e3 = mulmi(sub(d1, d3));
    const cplx X[8] = {add(c0, c1), add(e0, e1), add(c2, c3), add(e2, e3), sub(c0, c1), sub(e0, e1), sub(c2, c3), sub(e2, e3)};
    const cplx w1 = W[pp * s], w2 = W[2 * pp * s], w3 = W[3 * pp * s], w4 = W[4 * pp * s];
    const cplx w[8] = {cplx{1.0, 0.0}, w1, w2, w3, w4, cmul(w4, w1), cmul(w4, w2), cmul(w4, w3)};
    cplx *o = dst + row + q + 8 * s * pp;
    o[0] = X[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) o[k * s] = cmul(w[k], X[k]);
}

// A[i] = conj(A[i] * B[i]): the product, conjugated so that the FORWARD stages that follow compute the inverse transform
// (ifft(v) = conj(fft(conj(v))) / P; the consumer takes the conjugate and the 1 / P)
__global__ __launch_bounds__(256) void vadk_rsf_mulconj(cplx *A, const cplx *__restrict__ B, int64_t P, int64_t Pmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    cplx *a = A + (size_t)blockIdx.y * (size_t)Pmax + i;
    const cplx v = cmul(*a, B[i]);
    *a = {v.re, -v.im};
}

// spectrum edit between the two transforms: conv1 (conjugated, unscaled) -> a2[k] = Z[k] d[k] for k <= K, 0 up to P2
__global__ __launch_bounds__(256) void vadk_rsf_edit(const RsfParams p, const cplx *__restrict__ conv1, cplx *__restrict__ a2) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= p.P2) return;
    const size_t row = (size_t)blockIdx.y * (size_t)p.Pmax;
    cplx v = {0.0, 0.0};
    if (k <= p.K) {
        const cplx cv = conv1[row + k];
        const cplx conv = {cv.re / (double)p.P1, -cv.im / (double)p.P1};
        cplx X = cmul(chirp(k, p.n_in, -1.0), conv);
        double g = (k == 0 || (p.n_out % 2 == 0 && k == p.n_out / 2)) ? 1.0 : 2.0;
        const int64_t Np = p.n_in < p.n_out ? p.n_in : p.n_out;
        if (Np % 2 == 0 && k == Np / 2) g *= p.n_out < p.n_in ? 2.0 : (p.n_in < p.n_out ? 0.5 : 1.0);     // scipy's Nyquist rule
        X = {X.re * g, X.im * g};
        v = cmul(X, chirp(k, p.n_out, 1.0));
    }
    a2[row + k] = v;
}

// y[m] = Re(d[m] * conv2[m]) / N, conv2 = conj(stage output) / P2
__global__ __launch_bounds__(256) void vadk_rsf_store(const RsfParams p, const cplx *__restrict__ conv2) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= p.n_out) return;
    const cplx cv = conv2[(size_t)blockIdx.y * (size_t)p.Pmax + m];
    const cplx d = chirp(m, p.n_out, 1.0);
    const double re = d.re * cv.re + d.im * cv.im;                   // Re(d * conj(cv))
    p.y[(size_t)blockIdx.y * (size_t)p.n_out + m] = (float)(re / ((double)p.P2 * (double)p.n_in));
}

inline dim3 grid_for(int64_t n, int rows) { return dim3((unsigned)((n + 255) / 256), (unsigned)rows); }

// the stages of one FFT_P, ping-pong between u and v; returns the buffer that holds the result.  Radix 8 wherever three stages are
// left, then one radix-4 or radix-2 stage for the remainder: ceil(log2(P) / 3) passes over HBM.  -DRSF_RADIX=2 / 4 caps the radix
// (tools/variants.sh A/B).
#ifndef RSF_RADIX
#define RSF_RADIX 8
#endif
cplx *fft_stages(cplx *u, cplx *v, const cplx *W, int64_t P, int64_t Pmax, int rows, hipStream_t stream) {
    cplx *src = u, *dst = v;
    int64_t s = 1;
    int left = 0;                                   // stages (of radix 2) still to do
    while ((1ll << left) < P) ++left;
    while (left > 0) {
        // remainders first (left mod 3 at radix 8; left mod 2 at radix 4), so that the late stages - long contiguous runs - are the wide ones
        int r = 1;
        if (RSF_RADIX >= 8) r = left % 3 == 0 ? 3 : (left % 3 == 2 ? 2 : (left >= 4 ? 2 : 1));
        else if (RSF_RADIX >= 4) r = left % 2 == 0 ? 2 : 1;
        if (r == 3) hipLaunchKernelGGL(vadk_rsf_stage8, grid_for(P / 8, rows), dim3(256), 0, stream, src, dst, W, P, s, Pmax);
        else if (r == 2) hipLaunchKernelGGL(vadk_rsf_stage4, grid_for(P / 4, rows), dim3(256), 0, stream, src, dst, W, P, s, Pmax);
        else hipLaunchKernelGGL(vadk_rsf_stage, grid_for(P / 2, rows), dim3(256), 0, stream, src, dst, W, P, s, Pmax);
        s <<= r;
        left -= r;
        cplx *t = src; src = dst; dst = t;
    }
    return src;
}

}  // namespace

extern "C" hipError_t vadk_rsf_build_tables(const vadk::RsfParams *p, hipStream_t stream) {
    (void)hipGetLastError();
    cplx *W1 = reinterpret_cast<cplx *>(p->W1), *W2 = reinterpret_cast<cplx *>(p->W2);
    cplx *B1 = reinterpret_cast<cplx *>(p->B1), *B2 = reinterpret_cast<cplx *>(p->B2);
    cplx *tmp = reinterpret_cast<cplx *>(p->b);
    hipLaunchKernelGGL(vadk_rsf_twiddles, grid_for(p->P1 / 2, 1), dim3(256), 0, stream, W1, p->P1);
    hipLaunchKernelGGL(vadk_rsf_twiddles, grid_for(p->P2 / 2, 1), dim3(256), 0, stream, W2, p->P2);
    // forward: conj(c)[k - n] for k <= K, n < N: |k - n| < N;  inverse: conj(d)[m - k] for m < M, k <= K: -K <= m - k < M
    hipLaunchKernelGGL(vadk_rsf_kernel, grid_for(p->P1, 1), dim3(256), 0, stream, B1, p->P1, p->n_in, p->n_in, -1.0);
    cplx *r = fft_stages(B1, tmp, W1, p->P1, p->P1, 1, stream);
    if (r != B1 && hipMemcpyAsync(B1, r, sizeof(cplx) * (size_t)p->P1, hipMemcpyDeviceToDevice, stream) != hipSuccess) return hipGetLastError();
    const int64_t span2 = p->n_out > p->K + 1 ? p->n_out : p->K + 1;
    hipLaunchKernelGGL(vadk_rsf_kernel, grid_for(p->P2, 1), dim3(256), 0, stream, B2, p->P2, p->n_out, span2, 1.0);
    r = fft_stages(B2, tmp, W2, p->P2, p->P2, 1, stream);
    if (r != B2 && hipMemcpyAsync(B2, r, sizeof(cplx) * (size_t)p->P2, hipMemcpyDeviceToDevice, stream) != hipSuccess) return hipGetLastError();
    return hipGetLastError();
}

extern "C" hipError_t vadk_rsf_run(const vadk::RsfParams *p, hipStream_t stream) {
    (void)hipGetLastError();
    if (p->rows < 1 || p->rows > 65535) return hipErrorInvalidValue;
    cplx *a = reinterpret_cast<cplx *>(p->a), *b = reinterpret_cast<cplx *>(p->b);
    const cplx *W1 = reinterpret_cast<const cplx *>(p->W1), *W2 = reinterpret_cast<const cplx *>(p->W2);
    const cplx *B1 = reinterpret_cast<const cplx *>(p->B1), *B2 = reinterpret_cast<const cplx *>(p->B2);
    if (p->x_f64) hipLaunchKernelGGL(vadk_rsf_load<true>, grid_for(p->P1, p->rows), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(vadk_rsf_load<false>, grid_for(p->P1, p->rows), dim3(256), 0, stream, *p);
    cplx *r = fft_stages(a, b, W1, p->P1, p->Pmax, p->rows, stream);
    cplx *o = r == a ? b : a;
    hipLaunchKernelGGL(vadk_rsf_mulconj, grid_for(p->P1, p->rows), dim3(256), 0, stream, r, B1, p->P1, p->Pmax);
    r = fft_stages(r, o, W1, p->P1, p->Pmax, p->rows, stream);          // conj(ifft) * P1
    o = r == a ? b : a;
    hipLaunchKernelGGL(vadk_rsf_edit, grid_for(p->P2, p->rows), dim3(256), 0, stream, *p, r, o);
    cplx *r2 = fft_stages(o, r, W2, p->P2, p->Pmax, p->rows, stream);
    cplx *o2 = r2 == a ? b : a;
    hipLaunchKernelGGL(vadk_rsf_mulconj, grid_for(p->P2, p->rows), dim3(256), 0, stream, r2, B2, p->P2, p->Pmax);
    r2 = fft_stages(r2, o2, W2, p->P2, p->Pmax, p->rows, stream);
    hipLaunchKernelGGL(vadk_rsf_store, grid_for(p->n_out, p->rows), dim3(256), 0, stream, *p, r2);
    return hipGetLastError();
}

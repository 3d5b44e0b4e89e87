// Whole-array Fourier resampling for ANY (n_in -> n_out): what AudioUtils.resample_audio returns for every input it accepts
// (/root/reference/src/real_time_vad/utils/audio.py:39-49 -> scipy.signal.resample(x, int(len * ratio)), real input, no window).
//
// scipy's method (rfft, copy min(n_in, n_out) // 2 + 1 bins, fix the Nyquist bin, irfft, scale) is a linear operator whose
// entries depend on ONE number only, t = m / n_out - n / n_in:
//
//     y[m] = 1 / n_in * sum_n x[n] * D(t),      D(t) = sin(P pi t) / sin(pi t)  -  [corr] cos(2 pi Kmax t)
//
// with Kmax = min(n_in, n_out) // 2, P = 2 Kmax + 1 (every kept bin k >= 1 counts twice), and `corr` set when the last bin
// counts once: n_out >= n_in with n_in even (scipy halves X[n_in / 2]; for n_out == n_in irfft reads the Nyquist bin once).
// Downsampling to an even n_out doubles Y[n_out / 2], which irfft then reads once: weight two like the others.
//
// Both sines split over m and n (sin(A_m - B_n) = sin A_m cos B_n - cos A_m sin B_n), and cos(2 pi Kmax t) = (-1)^n cos(pi n_in m / n_out),
// so the operator is never stored: the kernel (csrc/resample_generic.hip) evaluates it in float64 from four numbers per n and
// five per m.  These tables are built here in float64 with the phases reduced in integers first.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

namespace vadk {

struct RsgTables {
    int64_t n_in = 0, n_out = 0;
    int64_t a = 0, b = 0, L = 0;     // n_in = g a, n_out = g b, L = g a b: t = ((m a - n b) mod L) / L
    int64_t P = 0;                   // 2 * (min(n_in, n_out) // 2) + 1
    int32_t corrected = 0;           // the last kept bin counts once
    std::vector<double> tn;          // [n_in][4]:  cos B_n, sin B_n, cos b_n, sin b_n      B_n = P pi n / n_in,  b_n = pi n / n_in
    std::vector<double> tm;          // [n_out][5]: sin A_m, cos A_m, sin a_m, cos a_m, C_m  A_m = P pi m / n_out, a_m = pi m / n_out,
                                     //             C_m = cos(pi n_in m / n_out)
};

// limits of one call: lengths below 2^31 and rows * n_in * n_out <= 2^42 operator entries (each is evaluated, none stored:
// measured 0.5e12 entries / s, so ~8 s of GPU time at the cap = 75 s of 48 kHz audio to 16 kHz, in launches of <= 2^34 entries)
constexpr int64_t RSG_MAX_LEN = (1ll << 31) - 1;
constexpr int64_t RSG_MAX_ENTRIES = 1ll << 42;

bool build_rsg_tables(int64_t n_in, int64_t n_out, RsgTables &out, std::string &err);

// D(t) / n_in for one (m, n) from the tables, on the host in float64 (tests without a GPU; the kernel does the same arithmetic)
double rsg_entry(const RsgTables &t, int64_t m, int64_t n);

struct RsgParams {
    const void *x;            // [rows][n_in] float32 or float64
    const double *tn, *tm;    // device copies of the tables
    double *partial;          // [nslice][rows][n_out]
    float *y;                 // [rows][n_out]
    int64_t n_in, n_out, a, b, L;
    int64_t m_begin, m_end;   // this launch's outputs
    int32_t rows, nslice, slice_len;
    int32_t x_f64, corrected;
    double peak;              // D at t == 0: P - corrected
    double inv_n_in;
};

// ---- long arrays: the same function as two chirp-z transforms on power-of-two float64 FFTs (csrc/resample_fft.hip) -------------
struct RsfParams {
    const void *x;            // [rows][n_in] float32 or float64
    float *y;                 // [rows][n_out]
    void *a, *b;              // work: [rows][Pmax] complex float64 each
    void *W1, *W2;            // twiddles e^{-2 pi i j / P}, j < P / 2
    void *B1, *B2;            // FFT_P of the wrapped chirp kernels of the forward / inverse transform
    int64_t n_in, n_out, K;   // K = min(n_in, n_out) / 2: the last bin scipy keeps
    int64_t P1, P2, Pmax;     // powers of two >= 2 n_in, >= 2 n_out; Pmax = the larger = row stride of a / b
    int32_t rows, x_f64;
};
// lengths the FFT path takes: P <= 2^26 (n <= 2^25 samples: 11 minutes of 48 kHz audio; 1 GB per work buffer and row)
constexpr int64_t RSF_MAX_LEN = 1ll << 25;

}  // namespace vadk

extern "C" hipError_t vadk_rsf_build_tables(const vadk::RsfParams *p, hipStream_t stream);
extern "C" hipError_t vadk_rsf_run(const vadk::RsfParams *p, hipStream_t stream);
extern "C" hipError_t vadk_launch_rsg_partial(const vadk::RsgParams *p, hipStream_t stream);
extern "C" hipError_t vadk_launch_rsg_finish(const vadk::RsgParams *p, hipStream_t stream);

// Host tables of the generic Fourier resampler (resample_generic.h): float64, phases reduced in integers.
#include "resample_generic.h"

#include <cmath>
#include <numeric>

namespace vadk {
namespace {

// sin and cos of pi * num / den (num >= 0, den > 0): num is reduced mod 2 den in integers, the rest to |angle| <= pi / 4,
// so the result is good to an ulp whatever the size of num
void sincos_pi_frac(unsigned __int128 num, int64_t den, double &s, double &c) {
    const int64_t r = (int64_t)(num % (unsigned __int128)(2 * (__int128)den));   // [0, 2 den)
    // quadrant: q = round(2 r / den) in 0..4, remainder f = r / den - q / 2 in [-1/4, 1/4]
    const int64_t q = (4 * (__int128)r + den) / (2 * (__int128)den);
    const double f = (double)(2 * (__int128)r - (__int128)q * den) / (2.0 * (double)den);
    const double pi = 3.14159265358979323846264338327950288;
    const double sf = std::sin(pi * f), cf = std::cos(pi * f);
    switch ((int)(q & 3)) {
        case 0: s = sf;  c = cf;  break;
        case 1: s = cf;  c = -sf; break;
        case 2: s = -sf; c = -cf; break;
        default: s = -cf; c = sf; break;
    }
}

}  // namespace

bool build_rsg_tables(int64_t n_in, int64_t n_out, RsgTables &t, std::string &err) {
    if (n_in < 1 || n_out < 1 || n_in > RSG_MAX_LEN || n_out > RSG_MAX_LEN) {
        err = "lengths must be in 1 .. 2^31 - 1";
        return false;
    }
    const int64_t g = std::gcd(n_in, n_out);
    t.n_in = n_in;
    t.n_out = n_out;
    t.a = n_in / g;
    t.b = n_out / g;
    t.L = t.a * n_out;
    const int64_t kmax = std::min(n_in, n_out) / 2;
    t.P = 2 * kmax + 1;
    t.corrected = (n_out >= n_in && n_in % 2 == 0) ? 1 : 0;
    t.tn.resize((size_t)n_in * 4);
    t.tm.resize((size_t)n_out * 5);
    for (int64_t n = 0; n < n_in; ++n) {
        double *o = &t.tn[(size_t)n * 4];
        sincos_pi_frac((unsigned __int128)t.P * (unsigned __int128)n, n_in, o[1], o[0]);
        sincos_pi_frac((unsigned __int128)n, n_in, o[3], o[2]);
    }
    for (int64_t m = 0; m < n_out; ++m) {
        double *o = &t.tm[(size_t)m * 5];
        sincos_pi_frac((unsigned __int128)t.P * (unsigned __int128)m, n_out, o[0], o[1]);
        sincos_pi_frac((unsigned __int128)m, n_out, o[2], o[3]);
        double s;
        sincos_pi_frac((unsigned __int128)n_in * (unsigned __int128)m, n_out, s, o[4]);
    }
    return true;
}

double rsg_entry(const RsgTables &t, int64_t m, int64_t n) {
    const double *tn = &t.tn[(size_t)n * 4], *tm = &t.tm[(size_t)m * 5];
    __int128 j = (__int128)m * t.a - (__int128)n * t.b;
    if (j < 0) j += t.L;
    double d;
    if (j == 0) d = (double)(t.P - t.corrected);
    else {
        const double num = tm[0] * tn[0] - tm[1] * tn[1];
        double den = tm[2] * tn[2] - tm[3] * tn[3];
        const int64_t jc = (int64_t)(2 * j > t.L ? j - t.L : j);
        const double r = (double)jc / (double)t.L;
        if (std::fabs(r) < 0x1p-10) {       // the difference of products loses its leading digits here: small-angle series of sin(pi r)
            const double w = 3.14159265358979323846264338327950288 * r, u = w * w;
            den = std::copysign(std::fabs(w * (1.0 - u / 6.0 * (1.0 - u / 20.0 * (1.0 - u / 42.0)))), den);
        }
        d = num / den;
        if (t.corrected) d -= (n & 1) ? -tm[4] : tm[4];
    }
    return d / (double)t.n_in;
}

}  // namespace vadk

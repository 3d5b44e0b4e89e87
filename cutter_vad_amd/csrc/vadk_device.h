// Device helpers shared by the fused model kernels (gfx950): MFMA fragment convention, quad stores,
// fenced scheduling, buffer-descriptor weight loads, fast activation functions.  See vad_layout.h.
#pragma once
#include <hip/hip_runtime.h>
#include "vad_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

namespace vadk { namespace dev {

__device__ __forceinline__ f32x16 mfma4(f32x4 w, f32x4 a, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a.w, acc, 0, 0, 0);
    return acc;
}

// accumulator initialised from 4 "lane-expanded" bias blocks (regs 4g..4g+3 <- block g)
__device__ __forceinline__ f32x16 acc_from(const f32x4 *ws) {
    f32x4 b0 = ws[0], b1 = ws[BLK_F4], b2 = ws[2 * BLK_F4], b3 = ws[3 * BLK_F4];
    f32x16 a;
    a.s0 = b0.x; a.s1 = b0.y; a.s2 = b0.z; a.s3 = b0.w;
    a.s4 = b1.x; a.s5 = b1.y; a.s6 = b1.z; a.s7 = b1.w;
    a.s8 = b2.x; a.s9 = b2.y; a.sa = b2.z; a.sb = b2.w;
    a.sc = b3.x; a.sd = b3.y; a.se = b3.z; a.sf = b3.w;
    return a;
}

// A quad stored as two 8-byte halves: values that come out of packed (pair) arithmetic or of scalar ops need not be copied
// into four consecutive registers first (the compiler merges the two halves into one ds_write2_b64)
__device__ __forceinline__ void st2(f32x4 *dst, f32x4 v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 *d = reinterpret_cast<f32x2 *>(dst);
    d[0] = f32x2{v.x, v.y};
    d[1] = f32x2{v.z, v.w};
}

// ReLU as ONE instruction: max of the BIT PATTERN, as a signed integer, with 0 - a float with its sign bit clear is a non-negative
// integer and is kept, one with the sign bit set (negative values, -0) is a negative integer and becomes +0.  fmaxf(v, 0) and
// v_med3_f32(v, 0, +inf) - which the compiler folds back into fmaxf - cost two v_max_f32 each, the first (v, v) only to quiet a
// signalling NaN the hardware would quiet anyway.  (Inline asm `v_max_f32 %0, 0, %1` is NOT an option: inline asm is opaque to the
// hazard recogniser, which then leaves out the wait states between an MFMA and a VALU read of its result - tried, Silero V4's
// probabilities came out wrong by 0.5.)
__device__ __forceinline__ float relu1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)}; }

__device__ __forceinline__ f32x4 quad_of(const f32x16 &a, int g) {
    switch (g) {
        case 0: return f32x4{a.s0, a.s1, a.s2, a.s3};
        case 1: return f32x4{a.s4, a.s5, a.s6, a.s7};
        case 2: return f32x4{a.s8, a.s9, a.sa, a.sb};
        default: return f32x4{a.sc, a.sd, a.se, a.sf};
    }
}

// write a 32-channel output tile (relu'd) as 8 quad rows starting at row `row0`;
// lane (m,h) owns quads row0 + 2g + h
__device__ __forceinline__ void store_tile_relu(f32x4 *region, int row0, int m, int h, const f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) region[(row0 + 2 * g + h) * QS + m] = relu4(quad_of(acc, g));
}

// v_exp_f32 / v_rcp_f32 / v_sqrt_f32 are 1-ulp hardware ops: |error| of sigmoid/tanh below 5e-7 absolute
__device__ __forceinline__ float sigmoidf_(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}
__device__ __forceinline__ float tanhf_(float v) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * v));
}
// one v_mul + one v_fma (written as an explicit fma: left as re*re + im*im the vectoriser pairs the two products into a
// v_pk_mul_f32 and pays three register copies per magnitude to line the operands up)
__device__ __forceinline__ float mag_(float re, float im) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(re, re, im * im)); }

__device__ __forceinline__ f32x16 acc_of(f32x4 b0, f32x4 b1, f32x4 b2, f32x4 b3) {
    f32x16 a;
    a.s0 = b0.x; a.s1 = b0.y; a.s2 = b0.z; a.s3 = b0.w;
    a.s4 = b1.x; a.s5 = b1.y; a.s6 = b1.z; a.s7 = b1.w;
    a.s8 = b2.x; a.s9 = b2.y; a.sa = b2.z; a.sb = b2.w;
    a.sc = b3.x; a.sd = b3.y; a.se = b3.z; a.sf = b3.w;
    return a;
}

// ---- quad / tile arithmetic as PACKED fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two lanes of a register pair per
// instruction, the scalar forms' rounding).  fp32 MFMAs and VALU instructions share the vector pipe on this chip, so every VALU
// instruction not issued is time; left to itself the compiler packs about a fifth of such code and scalarises every subtraction
// (the backend has no packed subtract: the packed add with its negate modifier is written out).  The add / mul helpers are
// compiled WITHOUT contraction: where a caller keeps a product and a sum apart (the loaders' folds: the int16 and float32
// instantiations must round alike) they stay apart after inlining.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
namespace pk {
__device__ __forceinline__ f32x4 cat(f32x2 lo, f32x2 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3); }
// a - b as ONE packed instruction.  The backend has no packed subtract (an fsub of a <2 x float> is split into two v_sub_f32, and
// fma(b, -1, a) is folded back into that fsub), so the -1 is made opaque: v_pk_fma_f32(b, k, a) with k = -1 from an empty asm - the
// product is exact, the one rounding is that of a - b.  NOT written as inline asm (`v_pk_add_f32 ... neg_lo:[0,1] neg_hi:[0,1]`, as
// it was): inline asm is opaque to the hazard recogniser, which then does not insert the wait states an MFMA result needs before a
// VALU instruction reads it - and this helper is applied to accumulators (enc0's Toom-3 interpolation).
__device__ __forceinline__ f32x2 sub2(f32x2 a, f32x2 b) {
    float k = -1.0f;
    asm("" : "+v"(k));
    return __builtin_elementwise_fma(b, f32x2{k, k}, a);
}
__device__ __forceinline__ f32x4 add(f32x4 a, f32x4 b) { return cat(a.lo + b.lo, a.hi + b.hi); }
__device__ __forceinline__ f32x4 sub(f32x4 a, f32x4 b) { return cat(sub2(a.lo, b.lo), sub2(a.hi, b.hi)); }
__device__ __forceinline__ f32x4 mul(f32x4 a, f32x4 b) { return cat(a.lo * b.lo, a.hi * b.hi); }
__device__ __forceinline__ f32x4 fma(f32x4 a, f32x4 b, f32x4 c) {
    return cat(__builtin_elementwise_fma(a.lo, b.lo, c.lo), __builtin_elementwise_fma(a.hi, b.hi, c.hi));
}
__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
// |re + i im| of a quad: fma(re, re, im * im) as in mag_(), packed, then the four square roots
__device__ __forceinline__ f32x4 mag(f32x4 re, f32x4 im) {
    const f32x4 q = fma(re, re, mul(im, im));
    return f32x4{__builtin_amdgcn_sqrtf(q.x), __builtin_amdgcn_sqrtf(q.y), __builtin_amdgcn_sqrtf(q.z), __builtin_amdgcn_sqrtf(q.w)};
}
// the same on a 32 x 32 tile's 16 accumulator registers, quad by quad
#define VADK_PK16(expr)                                                                                            \
    {                                                                                                              \
        f32x16 r_;                                                                                                 \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                         \
            const f32x4 q_ = (expr);                                                                               \
            switch (g_) {                                                                                          \
                case 0: r_.s0 = q_.x; r_.s1 = q_.y; r_.s2 = q_.z; r_.s3 = q_.w; break;                             \
                case 1: r_.s4 = q_.x; r_.s5 = q_.y; r_.s6 = q_.z; r_.s7 = q_.w; break;                             \
                case 2: r_.s8 = q_.x; r_.s9 = q_.y; r_.sa = q_.z; r_.sb = q_.w; break;                             \
                default: r_.sc = q_.x; r_.sd = q_.y; r_.se = q_.z; r_.sf = q_.w; break;                            \
            }                                                                                                      \
        }                                                                                                          \
        return r_;                                                                                                 \
    }
// sigmoidf_ / tanhf_ on a quad: the same operations per component (scale, v_exp_f32, 1 +, v_rcp_f32, and tanh's 1 - 2 r as one fma),
// the full-rate ones as packed instructions: half the issue slots of the LSTM cell's non-transcendental part
__device__ __forceinline__ f32x4 exp2_4(f32x4 v) {
    return f32x4{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y), __builtin_amdgcn_exp2f(v.z), __builtin_amdgcn_exp2f(v.w)};
}
__device__ __forceinline__ f32x4 rcp4(f32x4 v) {
    return f32x4{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y), __builtin_amdgcn_rcpf(v.z), __builtin_amdgcn_rcpf(v.w)};
}
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) { return rcp4(add(exp2_4(mul(v, splat(-1.44269504088896341f))), splat(1.0f))); }
__device__ __forceinline__ f32x4 tanh4(f32x4 v) {
    return fma(rcp4(add(exp2_4(mul(v, splat(2.88539008177792681f))), splat(1.0f))), splat(-2.0f), splat(1.0f));
}
__device__ __forceinline__ f32x4 q16(const f32x16 &a, int g) {
    return g == 0 ? f32x4{a.s0, a.s1, a.s2, a.s3} : g == 1 ? f32x4{a.s4, a.s5, a.s6, a.s7}
         : g == 2 ? f32x4{a.s8, a.s9, a.sa, a.sb} : f32x4{a.sc, a.sd, a.se, a.sf};
}
__device__ __forceinline__ f32x16 add16(const f32x16 &a, const f32x16 &b) VADK_PK16(add(q16(a, g_), q16(b, g_)))
__device__ __forceinline__ f32x16 sub16(const f32x16 &a, const f32x16 &b) VADK_PK16(sub(q16(a, g_), q16(b, g_)))
__device__ __forceinline__ f32x16 fma16(const f32x16 &a, float k, const f32x16 &c) VADK_PK16(fma(q16(a, g_), splat(k), q16(c, g_)))
__device__ __forceinline__ f32x16 mul16(const f32x16 &a, float k) VADK_PK16(mul(q16(a, g_), splat(k)))
#undef VADK_PK16
}  // namespace pk
#pragma clang fp contract(fast)

// No instruction may be moved across this point by the compiler's scheduler.  The kernel's
// software pipelining (weights for iteration i+1 are requested before the MFMAs of iteration i
// are issued) only survives hipcc's machine scheduler when it is fenced like this.
#define SB() __builtin_amdgcn_sched_barrier(0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// one 1 KiB weight block: 16 bytes per lane at byte offset voff = lane*16, block index in SGPRs
__device__ __forceinline__ f32x4 ldw(__amdgpu_buffer_rsrc_t rs, int voff, int blk) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, blk * 1024, 0));
}

// (float) s / d for an int16 s and d = 32767 or 32768, bit for bit the IEEE quotient that numpy's true division gives
// (vad_websocket_server.py:341): q = s * r with r = float(1 / d), then one Newton correction in two fmas.  Exhaustively equal over
// all 65 536 values of s (tools/i16_division_check.py, tests/test_host_logic.py); 3 instructions instead of the ~10 of the
// v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence - 64 samples per thread in the V5 kernel.
__device__ __forceinline__ float i16_div(int s, float d, float r) {
    const float x = (float)s;
    const float q = x * r;
    const float e = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ f32x4 gate4(f32x4 v, float thr) {
    // utils/audio.py:117-118: np.where(np.abs(x) > thr, x, 0.0); thr < 0 disables the gate
    // branch-free (a uniform branch here would split the caller's basic block and defeat its instruction interleave)
    const bool off = !(thr >= 0.f);
    v.x = ((fabsf(v.x) > thr) | off) ? v.x : 0.f;
    v.y = ((fabsf(v.y) > thr) | off) ? v.y : 0.f;
    v.z = ((fabsf(v.z) > thr) | off) ? v.z : 0.f;
    v.w = ((fabsf(v.w) > thr) | off) ? v.w : 0.f;
    return v;
}

} }  // namespace vadk::dev


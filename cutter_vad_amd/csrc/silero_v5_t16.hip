// Silero-VAD V5 (16 kHz) step kernel on 16-STREAM tiles, for small and medium batches (gfx950).
//
// silero_v5.hip carries 32 streams per workgroup, so a launch of n streams occupies n / 32 of the 256 CUs: 1 024 streams
// (BASELINE.json configs[1]) use 32 CUs, 4 096 (configs[3]) half the chip - and a tile takes the same ~47 us however few
// there are.  This kernel is the same network, the same algebra (frame ingest under the recurrent gate half, 4-way folded DFT,
// Toom-3 enc0, split-K enc2) and the same per-stream results, re-expressed on v_mfma_f32_16x16x4_f32: a workgroup (4 waves)
// carries 16 streams, so the same batch spreads over twice as many CUs and every MFMA / VALU phase is half as long.
// The engine picks it when a call has at most T16_MAX_STREAMS streams (engine.cpp).
//
// Fragment convention (v_mfma_f32_16x16x4_f32, D = A[16 x 4] B[4 x 16] + C): lane l = (n = l & 15, kq = l >> 4).
//   A: lane (row n, kq) holds W[row][k = kq]       B: lane (stream n, kq) holds X[k = kq][n]
//   D: lane (stream n, rq = kq) holds rows 4 rq .. 4 rq + 3 of the 16-row tile  ->  exactly one LDS quad.
// One k-iteration j contracts 16 channels: lane (n, kq) reads activation quad row 4 j + kq of stream n (ONE ds_read_b128, the B
// operands of 4 MFMAs: component i = channel 16 j + 4 kq + i) and a 1 KiB weight block gives it W[row n][16 j + 4 kq + i].
// A wave owns 32 output channels = two row tiles rt = 0, 1: channel 32 w + 16 rt + 4 rq + i lives in quad row 8 w + 4 rt + rq.
// LDS quad row: 16 streams x float4.  The folded STFT operands are written TRANSPOSED by the loader (16 lanes of a stream write
// 16 different rows), so that view of the memory has a padded row stride (QSL = 17 float4); every other tensor is written and
// read in the D / B fragment layout (a group of 16 lanes = 256 contiguous bytes) and uses the dense stride QSD = 16.  With the
// two views overlaid the workgroup needs 78.6 KB of LDS - under half a CU's 160 KB.
// Two workgroups per CU were measured (__launch_bounds__(256, 2): 256 registers, 14 spilled; 8 192 streams = 512 tiles):
// 51.8 us against 53.6 us for the two tiles of a CU one after the other and 49.5 us for the 32-stream tile kernel, and
// +1.2 us on every batch of <= 4 096 streams - the second wave of a SIMD only hides waits; fp32 MFMAs and VALU work share the
// vector datapath on this chip, also across waves.  So: one workgroup per CU, all 512 registers, and the engine uses this
// kernel for calls of at most 4 096 streams (256 tiles = one per CU) and the 32-stream tiles above that.
// Weight stream: pack_silero_v5_t16.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "vad_layout.h"
#include "sm_device.h"
#include "vadk_device.h"

using namespace vadk;

#ifdef VADK_STAMPS      // tools/kbench.sh -DKB_TILE16 -DVADK_STAMPS: the same phase indices as silero_v5.hip
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (lane == 0) P.stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + (k)] = clock64();          \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
using namespace vadk::dev;

namespace {

constexpr int MT16 = 16;
constexpr int QSL = 17;                   // loader view: rows 64 c + .. of the folded operands (po 0.. | qo 16.. | pe+ 32.. | pe- 40.. | qe- 48.. | qe+ 56..), 192 rows
constexpr int QSD = 16;                   // dense view of the same memory, used by everything else:
//   rows 0..159 Toom-3 planes (32 p + ch/4); the Nyquist channel takes 8 rows (values on the kq = 0 row of a group of four, zeros
//   on the other three): 160..163 = points 0, 1, -1, 2; 164..167 = infinity; rows 0..31 later enc1 output, then the LSTM input;
//   rows 168.. = enc0 output (168 + 32 c + ch/4), then the enc2 partials (168 + 16 half + ch/4); then h_{t-1} (32 rows)
constexpr int T_ROW_NYQ = 160;
constexpr int T_ROW_E = 168;
constexpr int T_ROWS_X = 264;
constexpr int T_ROWS_H = 32;
constexpr int T_FOLD_SINK = 192;          // loader view: 32 rows behind the three columns, where the loader lanes q >= 8 drop their duplicate quads
static_assert(T_ROWS_X * QSD >= (T_FOLD_SINK + 32) * QSL, "the loader view (and its sink rows) must end before h");
constexpr int T_LDS_F4 = (T_ROWS_X + T_ROWS_H) * QSD + 16 + 12 + 36 + 16 + 96 + 128;   // + head partials [4][16], |X128| [3][16], fold corrections [3][3][16], sink [64], state machines [16] x 96 B, gate biases [4 waves][4 gates][32 units]
constexpr int T_LSTM_BIAS_BLOCK = 8 + 64 + 64 + 2;   // block of a wave's LSTM section that holds its gate biases compact: floats [gate][unit]
static_assert(T_LDS_F4 * 16 <= 80 * 1024, "stays under half a CU's LDS");
// RS instantiation (fused resample -> step).  The resampler's folded input chunks (2 buffers x {ue, ve, uo, vo} x 16 quad rows,
// loader stride) are staged in the activation region, which is idle until the frame loop starts; the tile's 16 kHz frames
// F [16 streams][129 quads] (128 + 1 of padding: the recombination stores scalars down a column of streams) have 33 KB of their
// own behind the common layout (111.6 KB in all; the launch is one workgroup per CU anyway) - so a tile whose 16 streams come
// from TWO segments (the last streams of one input rate and the first of the next) can resample one part after the other, each
// with its own operator, without the second part's staging running over the first part's frames.
constexpr int FQ = 129;
constexpr int RS_CH_ROWS = 16;
constexpr int RS_BUF = 4 * RS_CH_ROWS * QSL;
static_assert(2 * RS_BUF <= 192 * QSL, "resampler staging must fit the loader view");
static_assert((T_LDS_F4 + MT16 * FQ) * 16 <= 160 * 1024, "RS: common layout + F must fit one CU");

__device__ __forceinline__ f32x4 mfma16(f32x4 w, f32x4 a, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc, 0, 0, 0);
    return acc;
}

}  // namespace

// A k-iteration's MFMAs with the NEXT iteration's requests issued in their shadow: ND LDS reads first, then one weight-block
// request behind every four MFMAs, NV times.  On this kernel's 32-cycle MFMAs a burst of 8 - 10 requests between two groups of 32
// costs ~10 % of the group (the 32-stream kernel's MFMAs are twice as long and were left alone, but for its STFT).
#define T16_IL(NV, ND)                                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x100, (ND), 0);                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < (NV); ++i_) {                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                      \
    }
// eight accumulators, the four components of a quad in turn: consecutive MFMAs are independent
#define T16_MMA8(G, WS, AV)                                                                                     \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_)                                                        \
            G[k_] = __builtin_amdgcn_mfma_f32_16x16x4f32(WS[k_][j_], (AV)[j_], G[k_], 0, 0, 0);

// RS: one tick for streams at other input rates (vad_step_rates): the tile first resamples its 16 chunks to 16 kHz into LDS -
// AudioUtils.resample_audio's Fourier method as the folded operator of resample.hip, on 16 x 16 x 4 tiles - and the frame loop
// ingests them from there: no second launch, no HBM round trip of the 16 kHz frames.  T = 1, float32 input.
// K8: the graph's 8 kHz sub-model on native 8 kHz audio in 256-sample frames (silero_v5.hip has the algebra: window 128, hop 64, three
// columns, 4-way fold with n = 1..31, 64 complex bins as four 16-row tiles - one per wave - bin 64 on the VALU, encoder.0 with 65
// input channels; from enc1 on the instantiations are the same): pools of at most 4 096 native-8 kHz sessions get one tile per CU
// like the 16 kHz model's (the 32-stream K8 kernel runs n / 32 CUs: 37 us whatever the size).  Loader: 8 lanes per stream, so a
// fold call covers TWO columns of the tile's 16 streams (threads 0..127 / 128..255).
// ONE: one frame per stream in the call (k_T == 1), instantiated without the frame loop (silero_v5.hip has the reasons); the fused
// resample -> step launch is always one frame.
template <bool F32IN, bool RS, bool K8 = false, bool ONE = false>
// One workgroup per CU also here.  Built for two (a tick's segments are padded to whole tiles, so it can have a few more tiles
// than CUs), the dispatcher packs consecutive workgroups onto the same CU: 258 tiles ran on ~130 CUs, 69.9 us per tick against
// 55.6 for the two-launch form - so the engine uses this launch only when the tick has at most one tile per CU.
// (the eight leading arguments: the fields of P the kernel needs first, preloaded into SGPRs - see silero_v5.hip; KP(f) = that copy)
__global__ void __launch_bounds__(NTHREADS, 1) silero_v5_step16(const float *k_wstream, float *k_state, SmSlot *k_sm, const int32_t *k_slots,
                                                                const void *k_frames, const int k_n, const uint32_t k_wstream_bytes, const int k_T,
                                                                const StepParams P, const RateParams R) {
#define KP(f) k_##f
    using namespace vadk::v5;
    static_assert(!RS || F32IN, "resampled frames are float32");
    static_assert(!(RS && K8), "the fused resampler feeds the 16 kHz model");
    constexpr int QL = K8 ? 8 : 16;               // loader lanes per stream = quads per quarter column
    constexpr int CS = 4 * QL;                    // folded-operand rows per column
    constexpr int PS = K8 ? 16 : 32;              // quad rows per Toom-3 plane
    constexpr int ROWN = K8 ? 80 : T_ROW_NYQ;     // the Nyquist channel's rows (8: values on the kq = 0 row of each group of four)
    constexpr int NJ0 = K8 ? 4 : 8;               // k-iterations of enc0
    __shared__ f32x4 lds[T_LDS_F4 + (RS ? MT16 * FQ : 0)];     // RS: + the tile's 16 kHz frames F (one workgroup per CU either way)
    f32x4 *const RX = lds;
    f32x4 *const RE = lds + T_ROW_E * QSD;
    f32x4 *const RH = lds + T_ROWS_X * QSD;
    float *const headp = reinterpret_cast<float *>(RH + T_ROWS_H * QSD);   // [4][16]
    float *const nyqv = headp + 64;              // [3][16]
    float *const fcor = nyqv + 48;               // [3 columns][y128, a64, b64][16 streams]
    constexpr int FCOR_SINK = 144;               // [64] floats after fcor
    SmSlot *const smL = reinterpret_cast<SmSlot *>(fcor + 144 + 64);
    f32x4 *const biasL = reinterpret_cast<f32x4 *>(smL + MT16);          // gate biases, compact: [4 waves][4 gates][8 quads of units]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15;                      // stream of this lane's MFMA column
    const int kq = lane >> 4;                     // channel group (B operand) = row quad of the D tile
    const int nq = kq * QSD + n;                  // lane's offset inside a group of 4 quad rows (dense view)
    const int nqL = kq * QSL + n;                 // the same in the loader view
    // RS: the tile carries the virtual streams 16 b .. 16 b + 15 of the segments laid end to end (vad_layout.h); which segment a
    // column belongs to is a handful of compares on kernel arguments
    const int vcol = (int)blockIdx.x * MT16 + n;
    int gf = vcol;
    bool live = vcol < KP(n);
    if constexpr (RS) {
        int s0 = R.seg[0].stream0, vs = 0;
#pragma unroll
        for (int k = 1; k < RATE_MAX_SEGS; ++k)
            if (k < R.nseg && vcol >= R.seg[k].vstart) { s0 = R.seg[k].stream0; vs = R.seg[k].vstart; }
        gf = s0 + vcol - vs;
        live = vcol < R.total;
    }
    const int tile0 = (int)blockIdx.x * MT16;                      // (not RS: the tile's first stream in the call's arrays)
    const int slot = live ? (KP(slots) ? KP(slots)[gf] : gf) : 0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(KP(wstream)), 0, (int)KP(wstream_bytes), 0x00020000);
    const int lane16 = lane * 16;
#define WL(blk) ldw(wrs, lane16, (blk))
    const int o_stft = (int)P.sect[w][S_STFT], o_nyq = (int)P.sect[w][S_NYQ], o_e0 = (int)P.sect[w][S_ENC0];
    const int o_e1 = (int)P.sect[w][S_ENC1], o_e2 = (int)P.sect[w][S_ENC2], o_e3 = (int)P.sect[w][S_ENC3];
    const int o_l = (int)P.sect[w][S_LSTM];
    const int T = (ONE || RS) ? 1 : KP(T);

    // ---- frame ingest set-up: 16 lanes per stream, 16 streams per fold call (ms = tid >> 4) ----
    const float thr = P.thresh;
    const int q = tid & (QL - 1);
    const bool q0 = q == 0;
    const int lms = K8 ? (tid >> 3) & 15 : tid >> 4;      // the loader's stream of this thread
    const int lcol = tid >> 7;                            // K8: which of a fold call's two columns
    constexpr bool f32in = F32IN;
    constexpr int qsh = f32in ? 4 : 3;
    const float sc = P.fmt == 1 ? 32767.0f : 32768.0f, rsc = 1.0f / sc;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(KP(frames)), 0, (int)((unsigned)KP(n) * (unsigned)T * ((f32in ? 2048u : 1024u) >> (K8 ? 1 : 0))), 0x00020000);
    u32x4 xa_[4], xb_[4], xc_[4];                  // raw quads of the three columns
    f32x4 *const F4 = lds + T_LDS_F4;              // RS: the tile's resampled frames
#define X_ISSUE(c, XR, tt)                                                                                      \
    if constexpr (RS) {                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                           \
            XR[k] = __builtin_bit_cast(u32x4, F4[(tid >> 4) * FQ + 32 * (c) + q + 16 * k]);                     \
    } else {                                                                                                    \
        const int fq = ((tile0 + (tid >> 4)) * T + (tt)) * 128 + 32 * (c) + q;                                  \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                           \
            XR[k] = __builtin_amdgcn_raw_buffer_load_b128(frs, (fq + 16 * k) << qsh, 0, 0);                     \
    }

    // 8 kHz: a frame is 64 quads, column c = quads 16 c .. 16 c + 31; lane q of a stream's 8 loads quads q, 8 + q, 16 + q, 24 + q
    // of column min(c0 + lcol, 2) (the second call's upper half repeats column 2: same values to the same places)
#define X_ISSUE8(c0, XR, tt)                                                                                    \
    {                                                                                                           \
        const int cc_ = (c0) + lcol < 2 ? (c0) + lcol : 2;                                                      \
        const int fq = ((tile0 + lms) * T + (tt)) * 64 + 16 * cc_ + q;                                          \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                           \
            XR[k] = __builtin_amdgcn_raw_buffer_load_b128(frs, (fq + 8 * k) << qsh, 0, 0);                      \
    }

    // ---- RS: the tile's parts.  A part = the columns c0 .. c1 - 1 of this tile that belong to one segment (one input rate); a tile
    //      at a rate boundary has two.
    struct Part {
        RateSeg S;
        int sk, c0, c1, ls0, Q, Kc, shape;
        bool valid;
    };
    constexpr int DEAD = 1 << 26;                  // an index (in quads / samples) past every buffer: loads return 0
    const int v0 = (int)blockIdx.x * MT16;
    auto mk_part = [&](int from) {
        Part pt{};
        pt.valid = false;
        if constexpr (RS) {
            for (int k = from; k < R.nseg; ++k) {                    // block-uniform: kernel arguments only
                const int a0 = max(R.seg[k].vstart, v0) - v0, a1 = min(R.seg[k].vstart + R.seg[k].n, v0 + MT16) - v0;
                if (a0 < a1) {
                    pt.S = R.seg[k];
                    pt.sk = k; pt.c0 = a0; pt.c1 = a1;
                    pt.ls0 = v0 - pt.S.vstart;                       // column c holds the segment's stream ls0 + c (c0 <= c < c1)
                    pt.Q = pt.S.n_in >> 2;
                    // the operator stream says how the part is contracted (pack_resample_operator_t16): 0 = every folded sample, two row
                    // tiles per wave; 1 / 2 = 48 / 24 kHz with every third sample copied ("P3"); 3 = 8 kHz with the even outputs copied
                    // and only the odd row tile contracted ("U2")
                    const int wbk = (int)pt.S.wave_blocks;
                    pt.shape = wbk == 4 + (pt.Q >> 4) * 8 ? 0 : wbk == 2 + (pt.Q >> 4) * 4 ? 3 : pt.S.n_in == 1536 ? 1 : 2;
                    pt.Kc = (pt.shape == 1 || pt.shape == 2) ? (2 * pt.Q) / 3 : pt.Q;      // contraction length per folded part
                    pt.valid = true;
                    break;
                }
            }
        }
        return pt;
    };
    // chunk loader: 16 streams x 16 folded quads = one per thread (stream ms = tid >> 4, quad ql = tid & 15)
    u32x4 xlA[6], xlB[6];
    const int cms = tid >> 4, cql = tid & 15;
    auto load_chunk = [&](auto p3tag, const Part &pt, int c, u32x4 *xl) {
        const __amdgpu_buffer_rsrc_t xrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pt.S.in), 0, (int)((unsigned)pt.S.n * (unsigned)pt.S.n_in * 4u), 0x00020000);
        const int Q = pt.Q, qq = cql + 16 * c, base = (cms >= pt.c0 && cms < pt.c1) ? (pt.ls0 + cms) * Q : DEAD;
        if constexpr (decltype(p3tag)::value == 1 || decltype(p3tag)::value == 2) {
            // P3 (24 / 48 kHz, see the part loop): compact quad g = the folded samples j = 6g+1, 6g+2, 6g+4, 6g+5; the thread reads
            // six consecutive samples of each of the four regions - x[6g ..], x[H + 6g ..], x[H - 6g - 6 ..], x[n - 6g - 6 ..] -
            // which hold those four j and the two samples 3 i' in between (8-byte aligned; wide loads need dword alignment only)
            const int s6 = 6 * qq, sb = base * 4, H = 2 * Q, nn = 4 * Q;
            xl[0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (sb + s6) * 4, 0, 0);
            xl[1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (sb + H + s6) * 4, 0, 0);
            xl[2] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (sb + H - s6 - 4) * 4, 0, 0);
            xl[3] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (sb + nn - s6 - 4) * 4, 0, 0);
            const u32x2 a2 = __builtin_amdgcn_raw_buffer_load_b64(xrs, (sb + s6 + 4) * 4, 0, 0);
            const u32x2 c2 = __builtin_amdgcn_raw_buffer_load_b64(xrs, (sb + H + s6 + 4) * 4, 0, 0);
            const u32x2 b2 = __builtin_amdgcn_raw_buffer_load_b64(xrs, (sb + H - s6 - 6) * 4, 0, 0);
            const u32x2 d2 = __builtin_amdgcn_raw_buffer_load_b64(xrs, (sb + nn - s6 - 6) * 4, 0, 0);
            xl[4] = u32x4{a2.x, a2.y, c2.x, c2.y};
            xl[5] = u32x4{b2.x, b2.y, d2.x, d2.y};
        } else {
            xl[0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + qq) * 16, 0, 0);
            xl[1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + qq + (Q >> 1)) * 16, 0, 0);
            xl[2] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (Q >> 1) - qq) * 16, 0, 0);
            xl[3] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (Q >> 1) - qq - 1) * 16, 0, 0);
            xl[4] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + (qq == 0 ? 0 : Q - qq)) * 16, 0, 0);
            xl[5] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (base + Q - qq - 1) * 16, 0, 0);
        }
    };
    Part cur = mk_part(0);
    // (requesting the first part's first chunk right here, IN FRONT of the state loads, was measured: 48.6 - 49.7 us against
    //  47.9 - 48.2 for 4 096 streams at 48 kHz on one box - it delays the state loads queued behind it.  Behind them: below.)

    // ---- prologue: h_{t-1} -> LDS quads (32 rows x 16 streams), c_{t-1} -> registers, state machines -> LDS ----
    // Request order = the order in which the frame loop needs things (vmcnt retires in issue order): h, the wave's gate biases
    // (compact: 128 floats, kept in LDS for the call) and the tile's state machines (their 16 threads only), then - not RS, where
    // a whole resampling phase sits in front of the frame loop - the frame loop's first requests: the W_hh blocks of its first two
    // groups and the frame's first two columns depend on kernel arguments only, and W_hh is the coldest part of the weight
    // stream; then the window and c.  Streams past n (the last tile's tail) read slot 0's state and compute on it: a stream is
    // a column of every MFMA, nothing crosses columns, and every store of the kernel is guarded by `live`.
    f32x4 hv[2];
    const int fm = tid & 15, part = tid >> 4;      // fm == n: ONE slot lookup serves h, c and the state machine
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) hv[qq] = reinterpret_cast<const f32x4 *>(KP(state) + (size_t)slot * 256)[part * 2 + qq];
    SB();
    auto bias2 = __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, (o_l + T_LSTM_BIAS_BLOCK) * 1024, 0);
    const bool sm_thread = (tid < MT16) && live;
    const int sm_slot = slot;
    f32x4 smq[6];
    if (tid < MT16) {
#pragma unroll
        for (int k = 0; k < 6; ++k) smq[k] = reinterpret_cast<const f32x4 *>(KP(sm) + slot)[k];
    }
    SB();
    f32x4 wA[8], wB[8];                            // W_hh blocks of a group, ping-pong
#define H_LDW(WS, g, WH) _Pragma("unroll") for (int k = 0; k < 8; ++k) WS[k] = WL((WH) + 8 * (g) + k);
#define H_FIRST(L, tt)                                                                                          \
    {                                                                                                           \
        H_LDW(wA, 0, (L) + 8 + 64)                                                                              \
        SB();                                                                                                   \
        H_LDW(wB, 1, (L) + 8 + 64)                                                                              \
        if constexpr (K8) { X_ISSUE8(0, xa_, tt) X_ISSUE8(2, xb_, tt) }                                         \
        else { X_ISSUE(0, xa_, tt) X_ISSUE(1, xb_, tt) }                                                        \
        SB();                                                                                                   \
        if constexpr (RS) { X_ISSUE(2, xc_, tt) SB(); }                                                         \
    }
    if constexpr (!RS) H_FIRST(o_l, 0)             // frames t > 0 request theirs at the end of frame t - 1; RS: at the top of the frame
    const f32x4 W1 = ldw(wrs, q * 16, o_nyq), W3 = ldw(wrs, (2 * QL + q) * 16, o_nyq);   // w[n], w[128 + n]  (8 kHz: w[64 + n])
    const float w64 = ldw(wrs, QL * 16, o_nyq).x;                                          // w[64]             (8 kHz: w[32])
    f32x4 cst[2];                                  // c of units 32 w + 16 rt + 4 kq + i
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) cst[rt] = *reinterpret_cast<const f32x4 *>(KP(state) + (size_t)slot * 256 + 128 + 32 * w + 16 * rt + 4 * kq);
    // RS, first part at 48 kHz (the tiles that set a mixed tick's time): its first chunk is requested right BEHIND the state loads -
    // loads return in order, so the state is not delayed - and has its HBM round trip under the state's waits, the LDS writes and
    // the operator prefetch.  Same box: 43.3 -> 42.9 us (4 096 streams at 48 kHz), configs[3] 46.0 -> 45.5; for 8 / 24 kHz first
    // parts it changes nothing (+- 0.1), so they keep the request where the part begins.
    bool chunk0_requested = false;
    if constexpr (RS) {
        if (cur.valid && cur.S.wstream != nullptr && cur.S.n_in == 1536) {
            if (cur.shape == 0) load_chunk(std::integral_constant<int, 0>{}, cur, 0, xlA);
            else load_chunk(std::integral_constant<int, 1>{}, cur, 0, xlA);
            chunk0_requested = true;
        }
    }
    SB();
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) RH[(part * 2 + qq) * QSD + fm] = hv[qq];
    reinterpret_cast<decltype(bias2) *>(biasL + 32 * w)[lane] = bias2;
    int seg_last = 0;
    if (tid < MT16) {
#pragma unroll
        for (int k = 0; k < 6; ++k) reinterpret_cast<f32x4 *>(smL + tid)[k] = smq[k];
    }
    const float hb = KP(wstream)[(size_t)P.sect[0][S_HEADB] * BLK_FLOATS];

    if constexpr (RS) {
        // ---- the tile's 16 chunks -> 512 samples at 16 kHz each, into F (resample.hip has the algebra; pack_resample_operator_t16
        //      the operator layout): four folded inputs ue / ve / uo / vo of length Q = n_in / 4 against four 128-row operators
        //      (se, ae, so, ao); wave w owns rows o = 32 w .. 32 w + 31 (two row tiles) of all four
        float *const Ff = reinterpret_cast<float *>(F4);
        while (cur.valid) {
        const RateSeg S = cur.S;
        const int c0 = cur.c0, c1 = cur.c1, ls0 = cur.ls0;
        auto in_part = [&](int c) { return c >= c0 && c < c1; };
        if (S.wstream == nullptr) {                                   // already 16 kHz (resample_audio returns its input): copy
            const __amdgpu_buffer_rsrc_t xrs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.in), 0, (int)((unsigned)S.n * 2048u), 0x00020000);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = it * NTHREADS + tid, ms = idx >> 7, qd = idx & 127;
                if (in_part(ms))
                    F4[ms * FQ + qd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ((ls0 + ms) * 128 + qd) * 16, 0, 0));
            }
        } else {
          auto run_part = [&](auto p3tag) {
            // P3 (24 / 48 kHz: n_in = 3 n'): the samples x[3 i'] sit ON output instants - R[o][3 i'] = (512 / n_in) [o == m i'] +
            // (-1)^(o - m i') / n_in with m = 1536 / n_in (pack_resample_operator_t16 checks it and leaves those columns out of the
            // stream) - so the loader threads copy them, scaled, straight to their place in F and add them into one alternating sum
            // per stream, and the MFMAs contract only the other two thirds of the folded samples: 4 chunks instead of 6 (48 kHz),
            // 2 instead of 3 (24 kHz).  x[0], x[H], x[Q], x[3Q] - the samples the fold cannot pair - are all of that kind.
            // U2 (8 kHz: up by two): every input sample IS an even output sample (R[2 i][i'] = [i == i'], checked by the packer), and
            // the parity of an output is the parity of its folded row - so the loader threads copy the chunk into the even places of
            // F, the stream holds ONE 16-row tile per wave (the odd rows 32 w + 2 r + 1) and half the MFMAs and the VALU rows go.
            // Compile-time per part shape (tag 0: every sample contracted, 1: 48 kHz, 2: 24 kHz, 3: 8 kHz - four instantiations of
            // this body): branches inside the chunk loop cost ~0.4 us per chunk.
            constexpr int SHAPE = decltype(p3tag)::value;
            constexpr bool P3 = SHAPE == 1 || SHAPE == 2, U2 = SHAPE == 3;
            constexpr int NB = U2 ? 4 : 8;                            // operator blocks per k-iteration
            constexpr int om = SHAPE;                                 // P3: output steps between two copied samples = 1536 / n_in
            const int Q = S.n_in >> 2, Kc = cur.Kc, nchunks = Kc >> 6;
            const float sc0 = 512.0f / (float)S.n_in;
            constexpr float sg3 = om == 1 ? -1.f : 1.f;
            float pA = 0.f;                                           // this thread's share of sum_i' (-1)^(m i') x[3 i']
            const __amdgpu_buffer_rsrc_t ors =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.wstream), 0, (int)S.wstream_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t xrs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(S.in), 0, (int)((unsigned)S.n * (unsigned)S.n_in * 4u), 0x00020000);
#define OL(blk) ldw(ors, lane16, (blk))
            const int wbase = w * (int)S.wave_blocks;
            if (!chunk0_requested) load_chunk(p3tag, cur, 0, xlA);
            chunk0_requested = false;
            // (input chunks are requested TWO chunks ahead - two register sets - so that a chunk's HBM round trip has a whole
            // chunk of MFMAs, ~1.7 us, more to hide under than it needs; chunk 0 has been on its way since before this part began)
            auto store_chunk = [&](int c, int buf, const u32x4 *xl) {
                if constexpr (P3) {
                    const f32x4 A4 = __builtin_bit_cast(f32x4, xl[0]), C4 = __builtin_bit_cast(f32x4, xl[1]);
                    const f32x4 B4 = __builtin_bit_cast(f32x4, xl[2]), D4 = __builtin_bit_cast(f32x4, xl[3]);
                    const f32x4 AC = __builtin_bit_cast(f32x4, xl[4]), BD = __builtin_bit_cast(f32x4, xl[5]);
                    // j = 6g+1, 6g+2, 6g+4, 6g+5: x[j], x[j+H] ascending; x[H-j], x[n-j] descending through their windows
                    const f32x4 a = f32x4{A4.y, A4.z, AC.x, AC.y}, cc = f32x4{C4.y, C4.z, AC.z, AC.w};
                    const f32x4 b = f32x4{B4.w, B4.z, B4.x, BD.y}, d = f32x4{D4.w, D4.z, D4.x, BD.w};
                    const f32x4 pe = a + cc, me = a - cc, qe = b + d, qo = b - d;
                    f32x4 *dst = lds + buf * RS_BUF + cql * QSL + cms;
                    dst[0] = pe + qe;
                    dst[RS_CH_ROWS * QSL] = pe - qe;
                    dst[2 * RS_CH_ROWS * QSL] = me + qo;
                    dst[3 * RS_CH_ROWS * QSL] = me - qo;
                    // the copied samples: x[6g], x[6g+3] (and the same past H) -> outputs ob, ob + m; x[H-6g-3], x[H-6g-6] -> 256 - ob - m,
                    // 256 - ob - 2m (likewise below 512): every output instant m i' exactly once over the part's chunks
                    const float e0 = A4.x + C4.x + BD.x + BD.z, e3 = A4.w + C4.w + B4.y + D4.y;
                    pA += e0 + sg3 * e3;
                    if (cms >= c0 && cms < c1) {
                        float *o = Ff + cms * (4 * FQ);
                        const int ob = om * 2 * (cql + 16 * c);
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        if constexpr (om == 1) {          // 48 kHz: neighbours - four aligned pairs
                            *reinterpret_cast<f32x2 *>(o + ob) = f32x2{sc0 * A4.x, sc0 * A4.w};
                            *reinterpret_cast<f32x2 *>(o + 256 + ob) = f32x2{sc0 * C4.x, sc0 * C4.w};
                            *reinterpret_cast<f32x2 *>(o + 254 - ob) = f32x2{sc0 * BD.x, sc0 * B4.y};
                            *reinterpret_cast<f32x2 *>(o + 510 - ob) = f32x2{sc0 * BD.z, sc0 * D4.y};
                        } else {                          // 24 kHz: every other output; the odd ones get nothing but the alternation
                            *reinterpret_cast<f32x4 *>(o + ob) = f32x4{sc0 * A4.x, 0.f, sc0 * A4.w, 0.f};
                            *reinterpret_cast<f32x4 *>(o + 256 + ob) = f32x4{sc0 * C4.x, 0.f, sc0 * C4.w, 0.f};
                            *reinterpret_cast<f32x4 *>(o + 252 - ob) = f32x4{sc0 * BD.x, 0.f, sc0 * B4.y, 0.f};
                            *reinterpret_cast<f32x4 *>(o + 508 - ob) = f32x4{sc0 * BD.z, 0.f, sc0 * D4.y, 0.f};
                        }
                    }
                } else {
                    const f32x4 a = __builtin_bit_cast(f32x4, xl[0]), cc = __builtin_bit_cast(f32x4, xl[1]);
                    const f32x4 b0 = __builtin_bit_cast(f32x4, xl[2]), b1 = __builtin_bit_cast(f32x4, xl[3]);
                    const f32x4 d0 = __builtin_bit_cast(f32x4, xl[4]), d1 = __builtin_bit_cast(f32x4, xl[5]);
                    const f32x4 b = f32x4{b0.x, b1.w, b1.z, b1.y}, d = f32x4{d0.x, d1.w, d1.z, d1.y};
                    const f32x4 pe = a + cc, me = a - cc, qe = b + d, qo = b - d;
                    f32x4 ue = pe + qe, ve = pe - qe, uo = me + qo, vo = me - qo;
                    if (cql + 16 * c == 0) { ue.x = pe.x; ve.x = 0.f; uo.x = 0.f; vo.x = me.x; }     // j = 0 has no partner
                    f32x4 *dst = lds + buf * RS_BUF + cql * QSL + cms;
                    dst[0] = ue;
                    dst[RS_CH_ROWS * QSL] = ve;
                    dst[2 * RS_CH_ROWS * QSL] = uo;
                    dst[3 * RS_CH_ROWS * QSL] = vo;
                    if constexpr (U2) {
                        // y[2 i] = x[i]: the four aligned quads this thread holds - x[4 qq ..], x[H + 4 qq ..], x[H - 4 qq - 4 ..],
                        // x[n - 4 qq - 4 ..] - cover every sample of the chunk exactly once over the 16 loader threads of a stream;
                        // the odd places get zeros here and their values from the recombination, behind the barrier
                        if (cms >= c0 && cms < c1) {
                            float *o = Ff + cms * (4 * FQ);
                            const int qq = cql + 16 * c;
                            auto put = [&](int first, f32x4 v) {
                                *reinterpret_cast<f32x4 *>(o + 2 * first) = f32x4{v.x, 0.f, v.y, 0.f};
                                *reinterpret_cast<f32x4 *>(o + 2 * first + 4) = f32x4{v.z, 0.f, v.w, 0.f};
                            };
                            put(4 * qq, a);
                            put(2 * Q + 4 * qq, cc);
                            put(2 * Q - 4 * qq - 4, b1);
                            put(4 * Q - 4 * qq - 4, d1);
                        }
                    }
                }
            };
            // the sample each half-size product cannot pair, x[Q] +- x[Q + H], is a rank-1 term (accumulator init, rows 128 / 384)
            const int sb = in_part(n) ? (ls0 + n) * S.n_in : (DEAD << 2);         // columns of other segments contract zeros
            float xa = 0.f, xb = 0.f;
            f32x4 mid = f32x4{0.f, 0.f, 0.f, 0.f}, ini[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ini[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!P3) {                                                  // (P3: x[Q] and x[3Q] are copied samples)
                xa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (sb + Q) * 4, 0, 0));
                xb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (sb + 3 * Q) * 4, 0, 0));
                if constexpr (U2) {                                               // (rows 128 / 384 are even: copies)
                    ini[0] = OL(wbase);
                    ini[1] = OL(wbase + 1);
                } else {
                    mid = ldw(ors, (Q >> 1) * 16, (int)S.row128_block);           // floats 2Q, 2Q + 1: RE[128][Q] / 2, RO[128][Q] / 2
#pragma unroll
                    for (int k = 0; k < 4; ++k) ini[k] = OL(wbase + k);
                }
            }
            // output rows 128 / 384 on the VALU: thread = (stream tid & 15, part tid >> 4); parts 0..7 dot ue with GSE[128],
            // 8..15 uo with GSO[128], two quads of every chunk each
            float r128 = 0.f;
            const int rpart = tid >> 4, pr = rpart & 7, psel = rpart >> 3;
#ifndef RS_D
#define RS_D 4
#endif
            constexpr int D = RS_D;                                   // operator blocks run D - 1 k-iterations ahead, across chunks (4 or 8: the slot of
                                                                      // a k-iteration must be static inside the two-chunk loop body)
            f32x4 wq[D][NB], xqA[4], xqB[4];
            int ws = wbase + (U2 ? 2 : 4);
#define R_LDW0(slot, j) _Pragma("unroll") for (int k = 0; k < NB; ++k) wq[slot][k] = OL(ws + NB * (j) + k);
            // (tools/variants.sh experiments, never defined in the product build: RS_EXP_NOLDW = the operator is not streamed,
            //  RS_EXP_NOMFMA = 4 VALU FMAs stand in for each group of 4 MFMAs, RS_EXP_NOX = the input chunks are loaded once)
#ifdef RS_EXP_NOLDW
#define R_LDW(slot, j)
#else
#define R_LDW(slot, j) R_LDW0(slot, j)
#endif
#ifdef RS_EXP_NOMFMA
#define RS_MMA(W, X, A) ((A) + (W) * (X))
#else
#define RS_MMA(W, X, A) mfma16((W), (X), (A))
#endif
#ifdef RS_EXP_NOLDW
#pragma unroll
            for (int d = 0; d < D; ++d) { R_LDW0(d, d % (D - 1)) }
#endif
#pragma unroll
            for (int d = 0; d < D - 1; ++d) { R_LDW0(d, d) }
            if (nchunks > 1) load_chunk(p3tag, cur, 1, xlB);
            f32x4 acc[NB];                                            // part p (se, ae, so, ao), row tile rt -> acc[2 p + rt]; U2: acc[p]
            if constexpr (U2) {
                acc[0] = ini[0] * (xa + xb);
                acc[2] = ini[1] * (xa - xb);
                acc[1] = acc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    acc[0 + rt] = ini[rt] * (xa + xb);
                    acc[4 + rt] = ini[2 + rt] * (xa - xb);
                    acc[2 + rt] = acc[6 + rt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            store_chunk(0, 0, xlA);
            __syncthreads();
            // Chunk c: the MFMAs read staging buffer c & 1.  Software pipeline inside a chunk: the activation quads of k-iteration
            // j + 1 are read from LDS BEFORE the MFMAs of j are issued (two register sets), chunk c + 1 (requested two chunks ago,
            // in `XS`) is folded into the other staging buffer while the MFMAs of j = 1 run, the VALU rows under j = 2, chunk c + 2
            // is requested into `XL` at the start - so that the chunk's only exposed LDS round trip is the first read behind
            // the barrier at its end.  (Before: four exposed reads + fold + barrier per chunk = 1.3 us on top of 1.7 us of MFMAs.)
#ifdef RS_EXP_NOX
#define RS_LOADX(c, XL)
#else
#define RS_LOADX(c, XL) load_chunk(p3tag, cur, (c), XL)
#endif
#define RS_XQ(XQ, X, j) _Pragma("unroll") for (int p4 = 0; p4 < 4; ++p4) XQ[p4] = (X)[(p4 * RS_CH_ROWS + 4 * (j)) * QSL + nqL];
#define RS_STEP(j, XC, XN, EXTRA)                                                                               \
                {                                                                                               \
                    R_LDW((4 * PAR_ + (j) + D - 1) % D, (j) + D - 1)   /* past j = 3: the next chunks' blocks (the stream is contiguous) */ \
                    if ((j) < 3) { RS_XQ(XN, X, (j) + 1) }                                                      \
                    EXTRA                                                                                       \
                    SB();                                                                                       \
                    _Pragma("unroll") for (int k = 0; k < NB; ++k) acc[k] = RS_MMA(wq[(4 * PAR_ + (j)) % D][k], XC[U2 ? k : k >> 1], acc[k]); \
                    SB();                                                                                       \
                }
#define RS_CHUNK(c, XS, XL, PAR)                                                                                \
            {                                                                                                   \
                constexpr int PAR_ = PAR;                                                                       \
                const f32x4 *X = lds + ((c) & 1) * RS_BUF;                                                      \
                asm volatile("" : "+s"(ws));                                                                    \
                f32x4 g128[2];                                                                                  \
                RS_STEP(0, xqA, xqB,                                                                            \
                        if ((c) + 2 < nchunks) RS_LOADX((c) + 2, XL);                                           \
                        if constexpr (!U2) {                                                                    \
                            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                       \
                                g128[i] = ldw(ors, (psel * (Kc >> 2) + 16 * (c) + 2 * pr + i) * 16, (int)S.row128_block); \
                        })                                                                                      \
                RS_STEP(1, xqB, xqA, if ((c) + 1 < nchunks) store_chunk((c) + 1, ((c) + 1) & 1, XS);)           \
                RS_STEP(2, xqA, xqB, if constexpr (!U2) {                                                       \
                    const int ms = tid & 15;                                                                    \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
                        const f32x4 uu = X[(psel * 2 * RS_CH_ROWS + 2 * pr + i) * QSL + ms];                    \
                        r128 += g128[i].x * uu.x + g128[i].y * uu.y + g128[i].z * uu.z + g128[i].w * uu.w;      \
                    }                                                                                           \
                })                                                                                              \
                RS_STEP(3, xqB, xqA, )                                                                          \
                ws += 4 * NB;                                                                                   \
                __syncthreads();                                                                                \
                if ((c) + 1 < nchunks) { RS_XQ(xqA, lds + (((c) + 1) & 1) * RS_BUF, 0) }                        \
            }
            RS_XQ(xqA, lds, 0)
            for (int c = 0; c < nchunks; c += 2) {
                RS_CHUNK(c, xlB, xlA, 0)
                if (c + 1 < nchunks) RS_CHUNK(c + 1, xlA, xlB, 1)
            }
#undef RS_CHUNK
#undef RS_STEP
#undef RS_XQ
#undef RS_LOADX
#undef RS_MMA
#undef R_LDW
#undef R_LDW0
            if constexpr (U2) {
                // the odd outputs: lane (n, kq) holds rows o = 32 w + 8 kq + 2 i + 1 of se, ae, so, ao.  The barrier that ended the
                // chunk is also the one that separates these scalars from the loader's zeros in the same places.
                const f32x4 se = acc[0], ae = acc[1], so = acc[2], ao = acc[3];
                const f32x4 pe = se + ae, me = se - ae, pO = so + ao, mO = so - ao;
                const f32x4 y0 = pe + pO, y1 = pe - pO, lo = me + mO, hi = me - mO;
                if (in_part(n)) {
                    float *o = Ff + n * (4 * FQ);
                    const int o0 = 32 * w + 8 * kq + 1;
                    o[o0] = y0.x;     o[o0 + 256] = y1.x; o[256 - o0] = lo.x; o[512 - o0] = hi.x;
                    o[o0 + 2] = y0.y; o[o0 + 258] = y1.y; o[254 - o0] = lo.y; o[510 - o0] = hi.y;
                    o[o0 + 4] = y0.z; o[o0 + 260] = y1.z; o[252 - o0] = lo.z; o[508 - o0] = hi.z;
                    o[o0 + 6] = y0.w; o[o0 + 262] = y1.w; o[250 - o0] = lo.w; o[506 - o0] = hi.w;
                }
            } else if constexpr (!P3) {
                // recombine: y[o] = se+ae+so+ao, y[o+256] = se+ae-so-ao, y[256-o] = se-ae+so-ao, y[512-o] = se-ae-so+ao
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int row = 32 * w + 16 * rt + 4 * kq;
                    const f32x4 se = acc[0 + rt], ae = acc[2 + rt], so = acc[4 + rt], ao = acc[6 + rt];
                    const f32x4 pe = se + ae, me = se - ae, pO = so + ao, mO = so - ao;
                    if (in_part(n)) {
                        F4[n * FQ + (row >> 2)] = pe + pO;
                        F4[n * FQ + 64 + (row >> 2)] = pe - pO;
                        const f32x4 lo = me + mO, hi = me - mO;
                        float *o = Ff + n * (4 * FQ);
                        if (row != 0) { o[256 - row] = lo.x; o[512 - row] = hi.x; }     // o = 0: y[256] and y[0] are written above
                        o[255 - row] = lo.y; o[511 - row] = hi.y;
                        o[254 - row] = lo.z; o[510 - row] = hi.z;
                        o[253 - row] = lo.w; o[509 - row] = hi.w;
                    }
                }
                headp[rpart * 16 + (tid & 15)] = r128;               // [16 parts][16 streams]: headp .. fcor are idle before the frame loop
                __syncthreads();
                if (tid < MT16 && in_part(tid)) {                     // tid < 16: this thread's MFMA column n is stream tid - xa / xb are its x[Q], x[3Q]
                    float e = 0.f, od = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { e += headp[k * 16 + tid]; od += headp[(8 + k) * 16 + tid]; }
                    e += mid.x * (xa + xb);
                    od += mid.y * (xa - xb);
                    Ff[tid * (4 * FQ) + 128] = e + od;
                    Ff[tid * (4 * FQ) + 384] = e - od;
                }
            } else {
                // rows 128 / 384 and the alternating sum meet in LDS first: the recombination needs the sum
                headp[rpart * 16 + (tid & 15)] = r128;
                float *const altL = fcor + FCOR_SINK;                 // [16 streams]
                float a = pA;
                a += __shfl_xor(a, 1, 16);
                a += __shfl_xor(a, 2, 16);
                a += __shfl_xor(a, 4, 16);
                a += __shfl_xor(a, 8, 16);
                if (cql == 0) altL[cms] = a * (1.0f / (float)S.n_in);
                __syncthreads();
                // the same recombination, on top of the copied samples already in F, plus (-1)^o times the stream's sum / n_in
                const float altv = altL[n];
                const f32x4 altq = f32x4{altv, -altv, altv, -altv};
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int row = 32 * w + 16 * rt + 4 * kq;
                    const f32x4 se = acc[0 + rt], ae = acc[2 + rt], so = acc[4 + rt], ao = acc[6 + rt];
                    const f32x4 pe = se + ae, me = se - ae, pO = so + ao, mO = so - ao;
                    if (in_part(n)) {
                        float *o = Ff + n * (4 * FQ);
                        F4[n * FQ + (row >> 2)] += pe + pO + altq;
                        F4[n * FQ + 64 + (row >> 2)] += pe - pO + altq;
                        const f32x4 lo = me + mO + altq, hi = me - mO + altq;       // 256 - row - k and 512 - row - k have the parity of k
                        if (row != 0) { o[256 - row] += lo.x; o[512 - row] += hi.x; }
                        o[255 - row] += lo.y; o[511 - row] += hi.y;
                        o[254 - row] += lo.z; o[510 - row] += hi.z;
                        o[253 - row] += lo.w; o[509 - row] += hi.w;
                    }
                }
                if (tid < MT16 && in_part(tid)) {                     // rows 128 / 384 (even): on top of the copied x[Q], x[3Q]
                    float e = altv, od = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { e += headp[k * 16 + tid]; od += headp[(8 + k) * 16 + tid]; }
                    Ff[tid * (4 * FQ) + 128] += e + od;
                    Ff[tid * (4 * FQ) + 384] += e - od;
                }
            }
          };
          if (cur.shape == 0) run_part(std::integral_constant<int, 0>{});
          else if (cur.shape == 1) run_part(std::integral_constant<int, 1>{});
          else if (cur.shape == 2) run_part(std::integral_constant<int, 2>{});
          else run_part(std::integral_constant<int, 3>{});
#undef OL
        }
        __syncthreads();                                              // this part of F is complete; staging and headp are free again
        cur = mk_part(cur.sk + 1);
        }                                                             // next part
    }

    STAMP(0);
    for (int t = 0;;) {                          // T >= 1; the back edge is at the bottom, behind the next frame's first requests
        int ws_stft = o_stft, ws_e0 = o_e0, ws_e1 = o_e1, ws_e2 = o_e2, ws_e3 = o_e3, ws_l = o_l;
        asm volatile("" : "+s"(ws_stft), "+s"(ws_e0), "+s"(ws_e1), "+s"(ws_e2), "+s"(ws_e3), "+s"(ws_l));
        // ---- recurrent gate half W_hh . h_{t-1} (8 k-iterations x {4 gates x 2 row tiles}) with the frame ingested under it ----
        f32x4 G[8];                               // gate q, row tile rt -> G[2 q + rt]
        {
            const int wh = ws_l + 8 + 64;
            auto decode = [&](u32x4 b) -> f32x4 {
                f32x4 v = __builtin_bit_cast(f32x4, b);
                if constexpr (!f32in) {
                    const int s0 = (int)(short)(b.x & 0xffffu), s1 = (int)(short)(b.x >> 16);
                    const int s2 = (int)(short)(b.y & 0xffffu), s3 = (int)(short)(b.y >> 16);
                    v = f32x4{i16_div(s0, sc, rsc), i16_div(s1, sc, rsc), i16_div(s2, sc, rsc), i16_div(s3, sc, rsc)};
                }
                return gate4(v, thr);
            };
            // (8 kHz: 8 lanes per stream - row_half_mirror, and lane 8 of a row, which row_shr:1 would feed from the neighbouring
            // stream's lane 7, takes `edge` by a select)
            auto mirror = [](float v) -> float {
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K8 ? 0x141 : 0x140, 0xf, 0xf, true));
            };
            auto shr1 = [&](float edge, float v) -> float {
                const float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
                return (K8 && q0) ? edge : r;
            };
            auto shl8 = [](float v) -> float {        // row_shl:8: lane q gets lane q + 8 (lanes 8..15: zero)
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x108, 0xf, 0xf, true));
            };
            // the fold of silero_v5.hip, one call per column: stream ms = tid >> 4, n = 4 q + j
#define X_FOLD(c, XR)                                                                                           \
    {                                                                                                           \
        _Pragma("clang fp contract(off)")                                                                       \
        const int ms = lms;                                                                                     \
        const f32x4 xA = decode(XR[0]), xB = decode(XR[1]), xC = decode(XR[2]), xD = decode(XR[3]);             \
        const float mBx = mirror(xB.x), mDx = mirror(xD.x);                                                     \
        const f32x4 y1 = pk::mul(xA, W1), y3 = pk::mul(xC, W3);                                                 \
        const f32x4 y2 = pk::mul(f32x4{shr1(xC.x, mBx), mirror(xB.w), mirror(xB.z), mirror(xB.y)}, W3);         \
        const f32x4 y4 = pk::mul(f32x4{shr1(0.f, mDx), mirror(xD.w), mirror(xD.z), mirror(xD.y)}, W1);          \
        const f32x4 s14 = pk::add(y1, y4), d14 = pk::sub(y1, y4), s23 = pk::add(y2, y3), d23 = pk::sub(y2, y3); \
        f32x4 pe = pk::add(s14, s23), po = pk::sub(s14, s23), qe = pk::sub(d14, d23), qo = pk::add(d14, d23);   \
        {                                                                                                       \
            pe.x = q0 ? 0.f : pe.x; po.x = q0 ? 0.f : po.x; qe.x = q0 ? 0.f : qe.x; qo.x = q0 ? 0.f : qo.x;        \
            const float y64 = xB.x * w64, y192 = xD.x * w64;                                                    \
            const int fo = q0 ? (c) * 48 + ms : FCOR_SINK + lane;                                               \
            fcor[fo] = y3.x;                                                                                    \
            fcor[fo + (q0 ? 16 : 0)] = y64 + y192;                                                              \
            fcor[fo + (q0 ? 32 : 0)] = y64 - y192;                                                              \
        }                                                                                                       \
        if constexpr (K8) {                                                                                     \
            st2(&RX[(CS * (c) + q) * QSL + ms], pe);                                                            \
            st2(&RX[(CS * (c) + QL + q) * QSL + ms], po);                                                       \
            st2(&RX[(CS * (c) + 2 * QL + q) * QSL + ms], qe);                                                   \
            st2(&RX[(CS * (c) + 3 * QL + q) * QSL + ms], qo);                                                   \
        } else {                                                                                                \
        /* the odd bins contract po | qo as they are; the even bins' operands fold once more, about n = 32 (vad_layout.h,            \
           bin_of_channel_fold3; silero_v5.hip has the lane algebra).  Lanes q < 8 hold n = 0..31 and store; lanes q >= 8 hold the \
           same values again and drop them into sink rows (a select on the address, no branch: the fold stays in the MFMAs' basic \
           block).  Slot n = 0 carries the unpaired n = 32 (lane 8, component 0): pe[32] | qe[32] */                  \
        const f32x4 pm = f32x4{shr1(0.f, mirror(pe.x)), mirror(pe.w), mirror(pe.z), mirror(pe.y)};               \
        const f32x4 qm = f32x4{shr1(0.f, mirror(qe.x)), mirror(qe.w), mirror(qe.z), mirror(qe.y)};               \
        f32x4 pep = pk::add(pe, pm), pen = pk::sub(pe, pm), qen = pk::sub(qe, qm), qep = pk::add(qe, qm);       \
        const float pe32 = shl8(pe.x), qe32 = shl8(qe.x);                                                       \
        pep.x = q0 ? pe32 : pep.x; pen.x = q0 ? 0.f : pen.x; qen.x = q0 ? 0.f : qen.x; qep.x = q0 ? qe32 : qep.x;   \
        st2(&RX[(64 * (c) + q) * QSL + ms], po);                                                               \
        st2(&RX[(64 * (c) + 16 + q) * QSL + ms], qo);                                                          \
        const int er = (q < 8 ? 64 * (c) + 32 + q : T_FOLD_SINK - 8 + q) * QSL + ms;                            \
        st2(&RX[er], pep);                                                                                      \
        st2(&RX[er + 8 * QSL], pen);                                                                            \
        st2(&RX[er + 16 * QSL], qen);                                                                           \
        st2(&RX[er + 24 * QSL], qep);                                                                           \
        }                                                                                                       \
    }
#define H_MMA(WS, g)                                                                                            \
    {                                                                                                           \
        const f32x4 av = RH[(4 * (g)) * QSD + nq];                                                             \
        T16_MMA8(G, WS, av)                                                                                     \
    }
#define H_MIX                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, 40, 0);                                                     \
    }
            if constexpr (RS) H_FIRST(ws_l, t)                 // F is dead once every wave has passed the barrier below
            {   // the accumulators start at the gate biases: G[2 q + rt] register i of a lane = unit 16 rt + 4 kq + i of gate q (the
                // 16 lanes of a row group read the same 16 bytes: a broadcast).  The wave reads what the wave itself wrote.
                const f32x4 *const bq = biasL + 32 * w + kq;
#pragma unroll
                for (int k = 0; k < 8; ++k) G[k] = bq[(k >> 1) * 8 + (k & 1) * 4];
            }
            __syncthreads();   // (0) h_{t-1} visible (t > 0: follows barrier (8))
            STAMP(31);
            H_MMA(wA, 0) SB();
            if constexpr (K8) {             // two fold calls: columns (0 | 1) by half of the workgroup, then column 2
            H_LDW(wA, 2, wh) H_MMA(wB, 1) T16_IL(8, 1) SB();
            H_LDW(wB, 3, wh) SB(); H_MMA(wA, 2) X_FOLD(lcol, xa_) H_MIX SB();
            H_LDW(wA, 4, wh) H_MMA(wB, 3) T16_IL(8, 1) SB();
            H_LDW(wB, 5, wh) SB(); H_MMA(wA, 4) X_FOLD(2, xb_) H_MIX SB();
            H_LDW(wA, 6, wh) H_MMA(wB, 5) T16_IL(8, 1) SB();
            H_LDW(wB, 7, wh) H_MMA(wA, 6) T16_IL(8, 1) SB();
            H_MMA(wB, 7) SB();
            } else {
            H_LDW(wA, 2, wh) if constexpr (!RS) { X_ISSUE(2, xc_, t) } SB(); H_MMA(wB, 1) SB();
            H_LDW(wB, 3, wh) SB(); H_MMA(wA, 2) X_FOLD(0, xa_) H_MIX SB();
            H_LDW(wA, 4, wh) H_MMA(wB, 3) T16_IL(8, 1) SB();
            H_LDW(wB, 5, wh) SB(); H_MMA(wA, 4) X_FOLD(1, xb_) H_MIX SB();
            H_LDW(wA, 6, wh) H_MMA(wB, 5) T16_IL(8, 1) SB();
            H_LDW(wB, 7, wh) SB(); H_MMA(wA, 6) X_FOLD(2, xc_) H_MIX SB();
            H_MMA(wB, 7) SB();
            }
#undef H_MIX
#undef H_MMA
#undef X_FOLD
        }
        f32x4 Sw[2];                              // STFT blocks of k-iteration 0: cos, -sin of the odd tile
#pragma unroll
        for (int k = 0; k < 2; ++k) Sw[k] = WL(ws_stft + k);
        SB();
        STAMP(1);
        __syncthreads();   // (1) folded x visible
        STAMP(2);

        // ---- bin 128 on the VALU: 48 (column, stream) pairs, 4 lanes each ----
        {
            const int pair = tid >> 2, pt = tid & 3;
            const int c = pair >> 4, ms = pair & 15;
            float a = 0.f;
            if (pair < 48) {
#pragma unroll
                // 16 kHz: sum_n pe[n] (-1)^n = the same sum over the pe+ rows (slot 0 = pe[32], sign +); 8 kHz: over its 8 pe rows
                for (int i = 0; i < 2; ++i) {
                    const f32x4 pp = RX[(CS * c + (K8 ? 0 : 32) + pt * 2 + i) * QSL + ms];
                    a += (pp.x - pp.y) + (pp.z - pp.w);
                }
            }
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            if (pair < 48 && pt == 0) nyqv[c * 16 + ms] = fabsf(a + fcor[(c * 3 + 0) * 16 + ms] + fcor[(c * 3 + 1) * 16 + ms]);
        }

        f32x4 e0b[2], E0w[10];
        if constexpr (K8) {
            // ---- STFT, 8 kHz sub-model: 64 complex bins = four 16-row tiles, ONE per wave (pack_dft4_wave_128_t16): wave w owns the
            //      bins 2 (16 (w & 1) + r) + (w >> 1), r = 0..15 - waves 0 / 1 the even bins (pe | qe), 2 / 3 the odd ones (po | qo);
            //      K = 32 = two k-iterations; the accumulators start from the rank-1 terms of n = 0, 32, 64 as in the 16 kHz model ----
            const bool even = w < 2;
            f32x4 sre[3], sim[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float y128 = fcor[(c * 3 + 0) * 16 + n], a64 = fcor[(c * 3 + 1) * 16 + n], b64 = fcor[(c * 3 + 2) * 16 + n];
                const float rp = even ? y128 + a64 : -y128, rm = even ? y128 - a64 : -y128;
                const float ip = even ? 0.f : -b64, im_ = even ? 0.f : b64;
                sre[c] = f32x4{rp, rm, rp, rm};
                sim[c] = f32x4{ip, im_, ip, im_};
            }
            const int rR = even ? 0 : QL, rI = even ? 2 * QL : 3 * QL;
            const f32x4 wr0 = Sw[0], wi0 = Sw[1], wr1 = WL(ws_stft + 2), wi1 = WL(ws_stft + 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 wr = j ? wr1 : wr0, wi = j ? wi1 : wi0;
                f32x4 u[3], v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    u[c] = RX[(CS * c + rR + 4 * j) * QSL + nqL];
                    v[c] = RX[(CS * c + rI + 4 * j) * QSL + nqL];
                }
                SB();
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    sre[c] = mfma16(wr, u[c], sre[c]);
                    sim[c] = mfma16(wi, v[c], sim[c]);
                }
                SB();
            }
            e0b[0] = WL(ws_e0); e0b[1] = WL(ws_e0 + 1);
#pragma unroll
            for (int k = 0; k < 10; ++k) E0w[k] = WL(ws_e0 + 2 + k);
            SB();
            __syncthreads();   // (1b) every wave is done reading the folded operands: the magnitudes may overwrite them
            {   // |.| -> Toom-3 evaluation planes, rows 16 p + channel / 4 = 16 p + 4 w + kq
                f32x4 mg[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) mg[c] = pk::mag(sre[c], sim[c]);
                const f32x4 s02 = pk::add(mg[0], mg[2]);
                f32x4 *o = RX + (4 * w) * QSD + nq;
                st2(o, mg[0]);
                st2(o + PS * QSD, pk::add(s02, mg[1]));
                st2(o + 2 * PS * QSD, pk::sub(s02, mg[1]));
                st2(o + 3 * PS * QSD, pk::fma(pk::splat(4.f), mg[2], pk::fma(pk::splat(2.f), mg[1], mg[0])));
                st2(o + 4 * PS * QSD, mg[2]);
            }
            if (tid < 64) {      // |X64|: values on the kq = 0 rows (80, 84), zeros on the other three of each group
                const float n0 = nyqv[n], n1 = nyqv[16 + n], n2 = nyqv[32 + n];
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                RX[ROWN * QSD + nq] = kq == 0 ? f32x4{n0, (n0 + n2) + n1, (n0 + n2) - n1, fmaf(4.f, n2, fmaf(2.f, n1, n0))} : z4;
                RX[(ROWN + 4) * QSD + nq] = kq == 0 ? f32x4{n2, 0.f, 0.f, 0.f} : z4;
            }
        } else
        // ---- STFT: wave w owns bins bin_of_channel_fold3(32 w + 16 rt + r): row tile 0 = 16 odd bins, cos on po, -sin on qo, K = 64
        //      (k-iterations 0..3); row tile 1 = 16 even bins on the once-more-folded operands pe+- | qe-+ (waves 2, 3 | 0, 1), K = 32
        //      (k-iterations 4, 5); 3 columns.  144 MFMAs per wave instead of 192 ----
        {
            f32x4 are[3][2], aim[3][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float y128 = fcor[(c * 3 + 0) * 16 + n], a64 = fcor[(c * 3 + 1) * 16 + n], b64 = fcor[(c * 3 + 2) * 16 + n];
                const float re1 = w < 2 ? y128 - a64 : y128 + a64;
                // register i of a D quad is tile row 4 rq + i: odd k: re = -y128, im = -+ b64 along the rows; even k = 2 m:
                // re = y128 + (-1)^m a64, im = 0
                are[c][0] = f32x4{-y128, -y128, -y128, -y128};
                aim[c][0] = f32x4{-b64, b64, -b64, b64};
                are[c][1] = f32x4{re1, re1, re1, re1};
                aim[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int eR = w < 2 ? 40 : 32, eI = w < 2 ? 56 : 48;
            const f32x4 *const XB = RX + nqL;                                                   // loader view
#define S_ROW_R(t) ((t) < 4 ? 4 * (t) : eR + 4 * ((t) - 4))
#define S_ROW_I(t) ((t) < 4 ? 16 + 4 * (t) : eI + 4 * ((t) - 4))
            f32x4 Aw[2], Bw[2], Au[3], Av[3], Bu[3], Bv[3];
#pragma unroll
            for (int k = 0; k < 2; ++k) Aw[k] = Sw[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) { Au[c] = XB[(64 * c + S_ROW_R(0)) * QSL]; Av[c] = XB[(64 * c + S_ROW_I(0)) * QSL]; }
#define S_LD(S, tt)                                                                        \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) S##w[k] = WL(ws_stft + 2 * (tt) + k);    \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) { S##u[c] = XB[(64 * c + S_ROW_R(tt)) * QSL]; S##v[c] = XB[(64 * c + S_ROW_I(tt)) * QSL]; }
#define S_MMA(S, rt)                                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                       \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                    \
            are[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(S##w[0][j_], S##u[c][j_], are[c][rt], 0, 0, 0); \
            aim[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(S##w[1][j_], S##v[c][j_], aim[c][rt], 0, 0, 0); \
        }
#define S_IL                                                                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
    }
            S_LD(B, 1) S_MMA(A, 0) S_IL SB();
            S_LD(A, 2) S_MMA(B, 0) S_IL SB();
            S_LD(B, 3) S_MMA(A, 0) S_IL SB();
            S_LD(A, 4) S_MMA(B, 0) S_IL SB();
            S_LD(B, 5) S_MMA(A, 1) S_IL SB();
            S_MMA(B, 1) SB();
#undef S_IL
#undef S_LD
#undef S_MMA
#undef S_ROW_R
#undef S_ROW_I
            e0b[0] = WL(ws_e0); e0b[1] = WL(ws_e0 + 1);
#pragma unroll
            for (int k = 0; k < 10; ++k) E0w[k] = WL(ws_e0 + 2 + k);
            SB();
            __syncthreads();   // (1b) every wave is done reading the folded operands: the magnitudes may overwrite them
            // |.| -> Toom-3 evaluation planes, rows 32 p + 8 w + 4 rt + kq
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x4 mg[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) mg[c] = pk::mag(are[c][rt], aim[c][rt]);
                const f32x4 s02 = pk::add(mg[0], mg[2]);
                f32x4 *o = RX + (8 * w + 4 * rt) * QSD + nq;
                st2(o, mg[0]);
                st2(o + 32 * QSD, pk::add(s02, mg[1]));
                st2(o + 64 * QSD, pk::sub(s02, mg[1]));
                st2(o + 96 * QSD, pk::fma(pk::splat(4.f), mg[2], pk::fma(pk::splat(2.f), mg[1], mg[0])));
                st2(o + 128 * QSD, mg[2]);
            }
            if (tid < 64) {      // |X128|: values on the kq = 0 rows (160, 164), zeros on the other three of each group
                const float n0 = nyqv[n], n1 = nyqv[16 + n], n2 = nyqv[32 + n];
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                RX[T_ROW_NYQ * QSD + nq] = kq == 0 ? f32x4{n0, (n0 + n2) + n1, (n0 + n2) - n1, fmaf(4.f, n2, fmaf(2.f, n1, n0))} : z4;
                RX[(T_ROW_NYQ + 4) * QSD + nq] = kq == 0 ? f32x4{n2, 0.f, 0.f, 0.f} : z4;
            }
        }
        STAMP(3);
        __syncthreads();   // (2) magnitudes complete
        STAMP(4);

        // ---- enc0 (Toom-3): five point-wise contractions over 128 channels (8 k-iterations) + the Nyquist channel ----
        f32x4 e1b[2], E1w[2];
        {
            const int ws = ws_e0 + 2;
            f32x4 acc[5][2];
#pragma unroll
            for (int p = 0; p < 5; ++p) acc[p][0] = acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 Aw[10], Bw[10], Aa[5], Ba[5];
#pragma unroll
            for (int k = 0; k < 10; ++k) Aw[k] = E0w[k];
#pragma unroll
            for (int p = 0; p < 5; ++p) Aa[p] = RX[(PS * p) * QSD + nq];
#define E0_LD(S, jj)                                                                       \
    _Pragma("unroll") for (int k = 0; k < 10; ++k) S##w[k] = WL(ws + 10 * (jj) + k);       \
    _Pragma("unroll") for (int p = 0; p < 5; ++p) S##a[p] = RX[(PS * p + 4 * (jj)) * QSD + nq];
#define E0_MMA(S)                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                       \
        _Pragma("unroll") for (int p = 0; p < 5; ++p) {                                    \
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(S##w[2 * p][j_], S##a[p][j_], acc[p][0], 0, 0, 0);         \
            acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(S##w[2 * p + 1][j_], S##a[p][j_], acc[p][1], 0, 0, 0);     \
        }
            for (int j = 0; j < NJ0; j += 2) {
                E0_LD(B, j + 1) E0_MMA(A) T16_IL(10, 5) SB();
                const int jn = j + 2 < NJ0 ? j + 2 : NJ0 - 2;
                E0_LD(A, jn) E0_MMA(B) T16_IL(10, 5) SB();
            }
#undef E0_LD
#undef E0_MMA
            {   // input channel 128 (Nyquist bin): K = 4 MFMAs whose k = 1..3 slots are zero on both operands
                const f32x4 an = RX[ROWN * QSD + nq], bn = RX[(ROWN + 4) * QSD + nq];
                const f32x4 wa0 = WL(ws + 10 * NJ0), wa1 = WL(ws + 10 * NJ0 + 1), wb0 = WL(ws + 10 * NJ0 + 2), wb1 = WL(ws + 10 * NJ0 + 3);
                e1b[0] = WL(ws_e1); e1b[1] = WL(ws_e1 + 1);
                E1w[0] = WL(ws_e1 + 2); E1w[1] = WL(ws_e1 + 3);
                SB();
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0.x, an.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1.x, an.x, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0.y, an.y, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1.y, an.y, acc[1][1], 0, 0, 0);
                acc[2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0.z, an.z, acc[2][0], 0, 0, 0);
                acc[2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1.z, an.z, acc[2][1], 0, 0, 0);
                acc[3][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0.w, an.w, acc[3][0], 0, 0, 0);
                acc[3][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1.w, an.w, acc[3][1], 0, 0, 0);
                acc[4][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb0.x, bn.x, acc[4][0], 0, 0, 0);
                acc[4][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb1.x, bn.x, acc[4][1], 0, 0, 0);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {      // interpolation (P(1), P(-1) arrive halved) + bias + ReLU -> rows 168 + 32 c + 8 w + 4 rt + kq
                const f32x4 bias = e0b[rt];
                const f32x4 y0 = acc[0][rt], y4 = acc[4][rt];
                const f32x4 bb = pk::sub(acc[1][rt], acc[2][rt]);
                const f32x4 y2 = pk::sub(pk::sub(pk::add(acc[1][rt], acc[2][rt]), y0), y4);
                const f32x4 t2 = pk::fma(y4, pk::splat(-16.0f), pk::fma(y2, pk::splat(-4.0f), pk::sub(acc[3][rt], y0)));
                const f32x4 y3 = pk::fma(bb, pk::splat(-1.0f / 3.0f), pk::mul(t2, pk::splat(1.0f / 6.0f)));
                f32x4 *o = RE + (8 * w + 4 * rt) * QSD + nq;
                o[0] = relu4(pk::add(pk::sub(bb, y3), bias));
                o[32 * QSD] = relu4(pk::add(y2, bias));
                o[64 * QSD] = relu4(pk::add(y3, bias));
            }
        }
        STAMP(5);
        __syncthreads();   // (3) enc0 out
        STAMP(6);

        // ---- enc1: 128 -> 64 ch, k3 s2 p1, 3 -> 2 columns; wave w: n-tile w & 1, column w >> 1; 16 k-iterations ----
        f32x4 e2b[2], E2w[2], e3b[2], E3w[2];
        {
            const int nt = w & 1, tp = w >> 1;
            const int ws = ws_e1 + 2;
            f32x4 acc[2] = {e1b[0], e1b[1]};
#define E1_ROW(it) (RE + ((tp + ((it) >> 3)) * 32 + 4 * ((it) & 7)) * QSD + nq)
            f32x4 Aw[2] = {E1w[0], E1w[1]}, Bw[2], Aa = *E1_ROW(0), Ba;
            for (int it = 0; it < 16; it += 2) {
                Bw[0] = WL(ws + 2 * (it + 1)); Bw[1] = WL(ws + 2 * (it + 1) + 1); Ba = *E1_ROW(it + 1); SB();
                acc[0] = mfma16(Aw[0], Aa, acc[0]); acc[1] = mfma16(Aw[1], Aa, acc[1]); SB();
                const int itn = it + 2 < 16 ? it + 2 : 14;
                Aw[0] = WL(ws + 2 * itn); Aw[1] = WL(ws + 2 * itn + 1); Aa = *E1_ROW(itn); SB();
                if (it == 14) {     // next layers' first blocks (enc2: this wave's K half)
                    const int ge = 2 + 8 * (w >> 1);
                    e2b[0] = WL(ws_e2); e2b[1] = WL(ws_e2 + 1);
                    E2w[0] = WL(ws_e2 + ge); E2w[1] = WL(ws_e2 + ge + 1);
                    e3b[0] = WL(ws_e3); e3b[1] = WL(ws_e3 + 1);
                    E3w[0] = WL(ws_e3 + 2); E3w[1] = WL(ws_e3 + 3);
                    SB();
                }
                acc[0] = mfma16(Bw[0], Ba, acc[0]); acc[1] = mfma16(Bw[1], Ba, acc[1]); SB();
            }
#undef E1_ROW
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) RX[(16 * tp + 8 * nt + 4 * rt) * QSD + nq] = relu4(acc[rt]);
        }
        STAMP(7);
        __syncthreads();   // (4) enc1 out in rows 0..31
        STAMP(8);

        // ---- enc2: 64 -> 64 ch, k3 s2 p1, 2 -> 1 column; split-K: wave w = tile w & 1, K half (= input column) w >> 1 ----
        {
            const int kh = w >> 1;
            const int ws = ws_e2 + 2 + 8 * kh;     // this half's 4 k-iterations x 2 row tiles
            f32x4 acc[2];
            acc[0] = kh == 0 ? e2b[0] : f32x4{0.f, 0.f, 0.f, 0.f};
            acc[1] = kh == 0 ? e2b[1] : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 wv[8], av[4];
            wv[0] = E2w[0]; wv[1] = E2w[1];
#pragma unroll
            for (int k = 2; k < 8; ++k) wv[k] = WL(ws + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = RX[(16 * kh + 4 * j) * QSD + nq];
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0] = mfma16(wv[2 * j], av[j], acc[0]); acc[1] = mfma16(wv[2 * j + 1], av[j], acc[1]); }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) RE[(16 * kh + 8 * (w & 1) + 4 * rt) * QSD + nq] = acc[rt];
        }
        STAMP(9);
        __syncthreads();   // (5) enc2 partials
        STAMP(10);

        // ---- enc3: 64 -> 128 ch, centre tap; input = relu(partial of K half 0 + K half 1) ----
        f32x4 Lw[8];
        {
            const int ws = ws_e3 + 2;
            f32x4 acc[2] = {e3b[0], e3b[1]};
            f32x4 wv[8], av[4];
            wv[0] = E3w[0]; wv[1] = E3w[1];
#pragma unroll
            for (int k = 2; k < 8; ++k) wv[k] = WL(ws + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = RE[(4 * j) * QSD + nq], b2 = RE[(16 + 4 * j) * QSD + nq];
                av[j] = relu4(f32x4{a.x + b2.x, a.y + b2.y, a.z + b2.z, a.w + b2.w});
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) Lw[k] = WL(ws_l + 8 + k);      // first k-iteration of the LSTM's input half
            SB();
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0] = mfma16(wv[2 * j], av[j], acc[0]); acc[1] = mfma16(wv[2 * j + 1], av[j], acc[1]); }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) RX[(8 * w + 4 * rt) * QSD + nq] = relu4(acc[rt]);
        }
        STAMP(11);
        __syncthreads();   // (6) LSTM input x in rows 0..31
        STAMP(12);

        // ---- LSTM: input half W_ih . x on top of the recurrent half, cell, head partial ----
        {
            const int ws = ws_l + 8;
            f32x4 Aw[8], Bw[8], Aa = RX[nq], Ba, hw[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) Aw[k] = Lw[k];
#define L_LD(S, it) _Pragma("unroll") for (int k = 0; k < 8; ++k) S##w[k] = WL(ws + 8 * (it) + k); S##a = RX[(4 * (it)) * QSD + nq];
#define L_MMA(S) T16_MMA8(G, S##w, S##a)
            for (int it = 0; it < 8; it += 2) {
                L_LD(B, it + 1) L_MMA(A) T16_IL(8, 1) SB();
                const int itn = it + 2 < 8 ? it + 2 : 6;
                if (it == 6) { hw[0] = WL(ws + 128); hw[1] = WL(ws + 129); SB(); }
                L_LD(A, itn) L_MMA(B) T16_IL(8, 1) SB();
            }
#undef L_LD
#undef L_MMA
            STAMP(18);
            STAMP(13);
            __syncthreads();   // (7) every wave is done reading h_{t-1}
            STAMP(14);
            f32x4 part4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 i4 = G[0 + rt], f4 = G[2 + rt], g4 = G[4 + rt], o4 = G[6 + rt], c4 = cst[rt], hwv = hw[rt];
                // c' = sigma(f) c + sigma(i) tanh(g); h' = sigma(o) tanh(c'); head partial += w relu(h') - a quad at a time, the
                // full-rate arithmetic packed (pk::), the transcendentals per component
                const f32x4 cn = pk::fma(pk::sigmoid4(f4), c4, pk::mul(pk::sigmoid4(i4), pk::tanh4(g4)));
                const f32x4 hn = pk::mul(pk::sigmoid4(o4), pk::tanh4(cn));
                part4 = pk::fma(hwv, relu4(hn), part4);
                RH[(8 * w + 4 * rt) * QSD + nq] = hn;
                if (t == T - 1 && live) {
                    *reinterpret_cast<f32x4 *>(KP(state) + (size_t)slot * 256 + 32 * w + 16 * rt + 4 * kq) = hn;
                    *reinterpret_cast<f32x4 *>(KP(state) + (size_t)slot * 256 + 128 + 32 * w + 16 * rt + 4 * kq) = cn;
                }
                cst[rt] = cn;
            }
            float part_ = (part4.x + part4.y) + (part4.z + part4.w);
            part_ += __shfl_xor(part_, 16);
            part_ += __shfl_xor(part_, 32);
            if (kq == 0) headp[w * 16 + n] = part_;
        }
        __syncthreads();   // (8) head partials + new h visible
        STAMP(15);

        if (tid < MT16) {
            const float z = hb + ((headp[tid] + headp[16 + tid]) + (headp[32 + tid] + headp[48 + tid]));
            const float p = fminf(sigmoidf_(z), 1.0f);
            if (sm_thread) {
                P.probs[(size_t)gf * T + t] = p;                       // tid < 16: this thread's column is stream gf
                SmSlot sm = smL[tid];
                int seg = 0;
                const int ev = sm_step(sm, p, &seg);
                if (t == T - 1) KP(sm)[sm_slot] = sm;
                else smL[tid] = sm;
                if (ev & 2) seg_last = seg;
                if (P.events) P.events[(size_t)gf * T + t] = (uint8_t)ev;
            }
        }
        if (++t >= T) break;
        if constexpr (!RS) H_FIRST(o_l, t)         // the next frame's first requests
    }
#undef H_FIRST
#undef H_LDW
#undef X_ISSUE
#undef WL
    if (sm_thread && P.seg_frames) P.seg_frames[gf] = seg_last;
}

#undef KP

#define V5_ARGS p->wstream, p->state, p->sm, p->slots, p->frames, (int)p->n, p->wstream_bytes, (int)p->T, *p
extern "C" hipError_t vadk_launch_silero_v5_t16(const vadk::StepParams *p, hipStream_t stream) {
    (void)hipGetLastError();
    const int tiles = (p->n + MT16 - 1) / MT16;
    if (tiles <= 0) return hipSuccess;
    const vadk::RateParams none{};
#define T16_LAUNCH(F, K, O) hipLaunchKernelGGL((silero_v5_step16<F, false, K, O>), dim3(tiles), dim3(vadk::NTHREADS), 0, stream, V5_ARGS, none)
    if (p->T < 1) return hipErrorInvalidValue;     // the frame loop tests its count at the bottom
    const bool k8 = p->variant != 0;               // the 8 kHz sub-model's blob (256-sample frames)
    const bool f32 = p->fmt == 0, one = p->T == 1;
    if (k8) {
        if (f32) { if (one) T16_LAUNCH(true, true, true); else T16_LAUNCH(true, true, false); }
        else { if (one) T16_LAUNCH(false, true, true); else T16_LAUNCH(false, true, false); }
    } else {
        if (f32) { if (one) T16_LAUNCH(true, false, true); else T16_LAUNCH(true, false, false); }
        else { if (one) T16_LAUNCH(false, false, true); else T16_LAUNCH(false, false, false); }
    }
#undef T16_LAUNCH
    return hipGetLastError();
}

// one tick for streams at several input rates: resample + step fused, one launch (p->T must be 1, p->n = all streams)
extern "C" hipError_t vadk_launch_silero_v5_t16_rates(const vadk::StepParams *p, const vadk::RateParams *r, hipStream_t stream) {
    (void)hipGetLastError();
    const int tiles = (r->total + MT16 - 1) / MT16;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL((silero_v5_step16<true, true>), dim3(tiles), dim3(vadk::NTHREADS), 0, stream, V5_ARGS, *r);
    return hipGetLastError();
}

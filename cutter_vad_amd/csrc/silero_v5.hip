// Fused Silero-VAD V5 (16 kHz) step kernel for MI355X / gfx950.
//
// Replaces `session.run` on silero_vad_v5.onnx's 16 kHz branch for a batch of independent
// streams (reference call site: /root/reference/src/real_time_vad/core/silero_model.py:433;
// dataflow: SURVEY.md §8 a7) together with the pre-steps the reference does in Python per
// frame: the denoise gate (utils/audio.py:117-118), the int16->float scaling
// (websocket_service/server/vad_websocket_server.py:341) and the hysteresis state machine
// (core/silero_model.py:790-949).
//
// One workgroup (4 waves) carries a tile of 32 streams through the WHOLE network:
//   recurrent gate half W_hh.h while the frame is loaded, gated, windowed and 4-way folded -> STFT as a folded DFT
//   (K = 64) -> |.| -> enc0 as a Toom-3 product (5 point-wise contractions) -> enc1 -> enc2 (split-K) -> enc3 ->
//   input gate half W_ih.x -> LSTM cell -> head -> sigmoid -> state machine.
// Activations never leave the CU (LDS quads, see vad_layout.h); weights stream from L2 into
// VGPRs in packed per-wave order; all contractions are v_mfma_f32_32x32x2_f32 (exact fp32).
// T > 1 frames per stream are processed in-kernel with h in LDS and c in registers.
// fp32 MFMAs and VALU instructions of a wave do NOT overlap on this chip (tools/ubench/mfma_valu.hip): the kernel's
// cost is MFMA cycles plus VALU cycles, which is why the algebra (folds, Toom-3) and the instruction counts matter.
#include <hip/hip_runtime.h>
#include "vad_layout.h"
#include "sm_device.h"
#include "vadk_device.h"

using namespace vadk;

#ifdef VADK_STAMPS
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (lane == 0) P.stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + (k)] = clock64();          \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

using namespace vadk::dev;

// F32IN: frames are float32 (else int16).  A template parameter, not a branch: the frame fold must stay in one basic
// block with the recurrent-half MFMAs for the instruction interleave below to be possible.
// K8: the graph's 8 kHz sub-model (If_0 else-branch) on 256-sample frames: the same dataflow at half the
// front-end size (vad_layout.h): window 128, hop 64, K = 32 per folded contraction, 64 complex bins as four 16-row tiles (one per
// wave, v_mfma_f32_16x16x4_f32), bin 64 on the VALU, encoder.0 with 65 input channels; everything from enc1 on is identical.
// The first eight arguments repeat the fields of P that the kernel needs before anything else (slot lookup, state, first weight
// blocks, first frame column): scalar arguments at the head of the list are PRELOADED into SGPRs by the command processor on
// gfx950 (`-mllvm -amdgpu-kernarg-preload-count=8` in _build.py: 13 dwords), struct fields are fetched with s_load after the wave
// has started.  Same box: 49.98 - 50.13 -> 49.84 - 49.87 us per step of 8 192 streams (KP(f) = that field's preloaded copy).
// ONE: the call steps one frame per stream (k_T == 1: the serving tick, vad_step, the bench) - instantiated WITHOUT the frame loop.
// With the loop the compiler hoists every loop-invariant address computation of the frame (~250 instructions, 94 of them only
// to park the result in an AGPR that the frame reads back once) in front of it, and keeps 100 more registers live; as straight-line
// code the addresses are formed where they are used: 2 137 -> 1 930 VALU instructions per wave, 369 -> 290 registers, and
// 46.35 -> 45.5 us per 8 192 streams on one box (profiles/r04_v5_single_frame_ab_same_box.log).  Same results, bit for bit.
template <bool F32IN, bool K8, bool ONE>
__global__ void __launch_bounds__(NTHREADS, 1) silero_v5_step(const float *k_wstream, float *k_state, SmSlot *k_sm, const int32_t *k_slots,
                                                              const void *k_frames, const int k_n, const uint32_t k_wstream_bytes, const int k_T,
                                                              const StepParams P) {
#define KP(f) k_##f
    using namespace vadk::v5;
    constexpr int QL = K8 ? 8 : 16;               // lanes per stream in the loader = quads per quarter column
    constexpr int CS = 4 * QL;                    // folded-operand rows per column (pe | po | qe | qo)
    constexpr int PS = K8 ? 16 : 32;              // quad rows per Toom-3 plane (|STFT| channels / 4)
    constexpr int ROWN = K8 ? ROW_NYQ_8K : ROW_NYQ;
    constexpr int NJ = K8 ? 4 : 8;                // k-iterations of the folded DFT
    constexpr int NJ0 = K8 ? 8 : 16;              // k-iterations of enc0
    __shared__ f32x4 lds[LDS_F4];
    f32x4 *const RX = lds;                       // the activation region (row map: vad_layout.h)
    f32x4 *const RE = lds + ROW_E * QS;          // its upper half
    f32x4 *const RH = lds + ROWS_X * QS;         // h
    float *const headp = reinterpret_cast<float *>(RH + ROWS_H * QS);   // [4][32]
    float *const nyqv = headp + 128;             // [3][32] |X128| per column
    float *const fcor = nyqv + 96;               // [3 columns][y128, a64, b64][32 streams]
    SmSlot *const smL = reinterpret_cast<SmSlot *>(fcor + 288 + 64);   // the tile's 32 state machines, resident for the call
    f32x4 *const biasL = reinterpret_cast<f32x4 *>(smL + MT);          // gate biases, compact: [4 waves][4 gates][8 quads of units]
    constexpr int FCOR_SINK = 288;               // [64] floats after fcor: where lanes q != 0 drop their (unused) correction terms

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;
    const int h = lane >> 5;
    const int tile0 = blockIdx.x * MT;
    STAMP(19);                                    // kernel entry
    const int gf = tile0 + m;                     // this lane's stream index within the call
    const bool live = gf < KP(n);
    const int slot = live ? (KP(slots) ? KP(slots)[gf] : gf) : 0;
    // Weight streams are read through ONE buffer descriptor (SGPRs, built from kernel arguments only):
    // voffset = lane * 16 (a single VGPR for every load of the kernel), soffset = block * 1024 (SALU).
    // With flat 64-bit VGPR addresses hipcc hoisted ~150 loop-invariant pointers out of the frame loop
    // and spilled them (cdna_hip_programming.md T8/T20).
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(KP(wstream)), 0, (int)KP(wstream_bytes), 0x00020000);
    const int lane16 = lane * 16;
#define WL(blk) ldw(wrs, lane16, (blk))
    const int o_stft = (int)P.sect[w][S_STFT], o_nyq = (int)P.sect[w][S_NYQ], o_e0 = (int)P.sect[w][S_ENC0];
    const int o_e1 = (int)P.sect[w][S_ENC1], o_e2 = (int)P.sect[w][S_ENC2], o_e3 = (int)P.sect[w][S_ENC3];
    const int o_l = (int)P.sect[w][S_LSTM];
    const int T = ONE ? 1 : KP(T);
    const int hq = h * QS + m;                    // lane's offset inside a quad-row pair

    // ---- frame ingest set-up (loop-invariant) ----
    const float thr = P.thresh;
    const int q = tid & (QL - 1);
    const bool q0 = q == 0;
    constexpr bool f32in = F32IN;
    constexpr int qsh = f32in ? 4 : 3;             // log2(bytes per 4-sample quad)
    const float sc = P.fmt == 1 ? 32767.0f : 32768.0f, rsc = 1.0f / sc;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(KP(frames)), 0, (int)((unsigned)KP(n) * (unsigned)T * ((f32in ? 2048u : 1024u) >> (K8 ? 1 : 0))), 0x00020000);
    u32x4 xa_[8], xb_[8];                          // raw quads of a column (both 16-stream halves), as bits
    // Column c of frame tt -> XR: lane q of a 16-lane row (one stream per row) loads quads q, 16+q, 32+q, 48+q of the
    // column: every sample once, 4 branch-free 16-byte loads per lane and stream half.
#define X_ISSUE(c, XR, tt)                                                                                      \
    _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) {                                                          \
        const int fq = ((tile0 + rr * 16 + (tid >> 4)) * T + (tt)) * 128 + 32 * (c) + q;   /* quad q of the column */ \
        /* ONE instruction stream for both formats: a quad is 16 bytes (f32) or 8 bytes (int16, the upper 8 bytes     \
           of the 16 loaded are ignored); a format branch here breaks the compiler's vmcnt bookkeeping */        \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                           \
            XR[rr * 4 + k] = __builtin_amdgcn_raw_buffer_load_b128(frs, (fq + 16 * k) << qsh, 0, 0);            \
    }
    // 8 kHz: a frame is 64 quads, column c = quads 16c .. 16c + 31; 8 lanes per stream, lane q loads quads q, 8+q, 16+q, 24+q
    // of the column into XR[XO .. XO + 3]; one call covers the tile's 32 streams
#define X_ISSUE8(c, XR, XO, tt)                                                                                 \
    {                                                                                                           \
        const int fq = ((tile0 + (tid >> 3)) * T + (tt)) * 64 + 16 * (c) + q;                                   \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                           \
            XR[(XO) + k] = __builtin_amdgcn_raw_buffer_load_b128(frs, (fq + 8 * k) << qsh, 0, 0);               \
    }

    // ---- prologue: h_{t-1} -> LDS quads, c_{t-1} -> registers (this lane's 16 units) -------
    // Request order = the order in which the frame loop needs things (vmcnt retires in issue order): h and the tile's state
    // machines (stored to LDS before barrier (0)), then the frame loop's first requests - the gate biases, the W_hh blocks of
    // its first two groups, the frame's first column(s): they depend on kernel arguments only, and W_hh is the coldest part of
    // the weight stream (last touched at the start of the previous launch, behind a launch's worth of frames and state in L2),
    // so its round trip now overlaps the state's instead of following it - then the window and c, which nothing needs before
    // the fold of group 2 and the cell.  Only h and the state machines are waited for before the first MFMA.
    // Streams past n (the last tile's tail) read slot 0's state - valid memory - and compute on it: a stream is a column of
    // every MFMA, nothing crosses columns, and every store of the kernel is guarded by `live`.
    f32x4 hv[4];
    const int fm = tid & 31, part = tid >> 5;
    // fm == m (a wave is 64 lanes): the stream this thread loads h for is the stream of its MFMA column -> ONE slot lookup
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) hv[qq] = reinterpret_cast<const f32x4 *>(KP(state) + (size_t)slot * 256)[part * 4 + qq];
    SB();                                          // h is on its way before anything waits for a section offset (a scalar load)
    // the wave's gate biases - the accumulators' initial values - compact: 128 floats [gate][unit], two per lane, kept in LDS
    // for the call and broadcast from there (lane-expanded blocks in the stream were 16 KB per wave at the head of the queue)
    auto bias2 = __builtin_amdgcn_raw_buffer_load_b64(wrs, lane * 8, (o_l + LSTM_BIAS_BLOCK) * 1024, 0);
    // the slot's state machine (96 B in HBM between calls) lives in LDS for the call and goes back with the last frame: its HBM
    // latency is off the tail of the kernel
    const bool sm_thread = (tid < MT) && live;
    const int sm_slot = slot;
    f32x4 smq[6];
    if (tid < MT) {
#pragma unroll
        for (int k = 0; k < 6; ++k) smq[k] = reinterpret_cast<const f32x4 *>(KP(sm) + slot)[k];
    }
    SB();
    f32x4 wA[8], wB[8];                            // W_hh blocks of a group of 2 k-iterations, ping-pong
#define H_LDW(WS, g, WH) _Pragma("unroll") for (int k = 0; k < 8; ++k) WS[k] = WL((WH) + 8 * (g) + k);
#define H_FIRST(L, tt)                                                                                          \
    {                                                                                                           \
        H_LDW(wA, 0, (L) + 16 + 64)                                                                             \
        SB();                                                                                                   \
        H_LDW(wB, 1, (L) + 16 + 64)                                                                             \
        if constexpr (K8) { X_ISSUE8(0, xa_, 0, tt) X_ISSUE8(1, xa_, 4, tt) }                                   \
        else { X_ISSUE(0, xa_, tt) }                                                                            \
        SB();                                                                                                   \
    }
    H_FIRST(o_l, 0)                                // frames t > 0 request theirs at the end of frame t - 1
    const f32x4 W1 = ldw(wrs, q * 16, o_nyq), W3 = ldw(wrs, (2 * QL + q) * 16, o_nyq);   // window w[n], w[128+n] = w[128-n]  (8 kHz: w[64+n])
    const float w64 = ldw(wrs, QL * 16, o_nyq).x;                                           // w[64] = w[192]                     (8 kHz: w[32] = w[96])
    f32x16 cst;   // c state of units 32w + 8g + 4h + i  (reg 4g+i)
    {
        f32x4 c4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) c4[g] = *reinterpret_cast<const f32x4 *>(KP(state) + (size_t)slot * 256 + 128 + 32 * w + 8 * g + 4 * h);
        cst = acc_of(c4[0], c4[1], c4[2], c4[3]);
    }
    SB();
    STAMP(29);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) RH[(part * 4 + qq) * QS + fm] = hv[qq];
    reinterpret_cast<decltype(bias2) *>(biasL + 32 * w)[lane] = bias2;
    STAMP(30);
    int seg_last = 0;
    if (tid < MT) {
#pragma unroll
        for (int k = 0; k < 6; ++k) reinterpret_cast<f32x4 *>(smL + tid)[k] = smq[k];
    }
    const float hb = KP(wstream)[(size_t)P.sect[0][S_HEADB] * BLK_FLOATS];    // head bias

    // every kernel argument the frame loop needs, fetched NOW (scalar loads from the kernarg segment cost a round
    // trip when they are left to the point of first use, after the barrier)
    {
        const void *a0 = KP(state), *a1 = KP(sm), *a2 = P.probs, *a3 = P.events, *a4 = KP(frames);
        const int a5 = __builtin_bit_cast(int, P.thresh), a6 = P.fmt;
        asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6));
    }
    STAMP(0);
    for (int t = 0;;) {                          // T >= 1 (the launcher never passes less); the back edge is at the bottom, behind the next frame's first requests
        // section offsets, made opaque per frame: otherwise every block offset of the kernel (~300 SGPR
        // values) is hoisted out of this loop as loop-invariant and spilled
        int ws_stft = o_stft, ws_nyq = o_nyq, ws_e0 = o_e0, ws_e1 = o_e1, ws_e2 = o_e2, ws_e3 = o_e3, ws_l = o_l;
        asm volatile("" : "+s"(ws_stft), "+s"(ws_nyq), "+s"(ws_e0), "+s"(ws_e1), "+s"(ws_e2), "+s"(ws_e3), "+s"(ws_l));
        // ---- the frame is ingested UNDER the recurrent half of the LSTM gates ----
        // W_hh . h_{t-1} does not depend on the frame, so its 256 MFMAs per wave run while the columns are in flight from
        // HBM and while they are folded.  vmcnt retires in issue order: the W_hh blocks of a group are requested BEFORE
        // any column requested in the same group, so waiting for a weight block never waits for the frame.
        // Fold: output (column c, stream ms, quad q), n = 4q..4q+3, from the column's quads q (y[n]), 32+q (y[128+n]),
        // 32-q / 31-q (y[128-n], reversed) and 64-q / 63-q (y[256-n], reversed); 16 lanes run over q.
        f32x16 gi, gfo, gg, go;
        {
            const int wh = ws_l + 16 + 64;                 // W_hh blocks: iteration it at wh + 4 it
            // The recurrent half runs as 8 groups of 2 k-iterations (32 MFMAs, W_hh blocks ping-ponged one group
            // ahead).  Columns 0, 1 and 2 are requested at groups 0, 1 and 4 (column 2 into column 0's buffer) and folded,
            // one 16-stream half per group, at groups 2..7.  Inside a group the fold is interleaved between the MFMAs
            // with sched_group_barrier: its LDS writes and the W_hh requests overlap the MFMAs; its VALU work does NOT
            // (tools/ubench/mfma_valu.hip: an fp32 MFMA and VALU instructions of the same wave serialise, 64 + 4.4 n
            // cycles), so the fold costs what its instruction count says.
            // int16 payloads are raw bits: decode (true division, like np.int16 -> float32 / 32767.0), then gate
            auto decode = [&](u32x4 b) -> f32x4 {
                f32x4 v = __builtin_bit_cast(f32x4, b);
                if constexpr (!f32in) {
                    const int s0 = (int)(short)(b.x & 0xffffu), s1 = (int)(short)(b.x >> 16);
                    const int s2 = (int)(short)(b.y & 0xffffu), s3 = (int)(short)(b.y >> 16);
                    v = f32x4{i16_div(s0, sc, rsc), i16_div(s1, sc, rsc), i16_div(s2, sc, rsc), i16_div(s3, sc, rsc)};
                }
                return gate4(v, thr);
            };
            // reversed reads come from the other lanes of the row: row_mirror hands lane q the value of lane 15 - q
            // (quad 31 - q of B, 63 - q of D); one more row_shr:1 gives quad 32 - q / 64 - q (lane 0 keeps `edge`)
            // (8 kHz: 8 lanes per stream - row_half_mirror, and lane 8 of a row, which row_shr:1 would feed from the
            // neighbouring stream's lane 7, takes `edge` by a select)
            auto mirror = [](float v) -> float {      // every lane of a row has a source: no `old` value to materialise
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K8 ? 0x141 : 0x140, 0xf, 0xf, true));
            };
            auto shl8 = [](float v) -> float {        // row_shl:8: lane q gets lane q + 8 (lanes 8..15: zero)
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x108, 0xf, 0xf, true));
            };
            auto shr1 = [&](float edge, float v) -> float {
                const float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
                return (K8 && q0) ? edge : r;
            };
            // Fold of (column c, stream half rr): n = 4q + j.  y1 = w[n] x[n] (A), y3 = w[128+n] x[128+n] (C),
            // y2 = w[128-n] x[128-n] (B reversed), y4 = w[256-n] x[256-n] (D reversed); w is symmetric about 128.
#define X_FOLD1(c, rr, XR) X_FOLDG(c, (rr) * 16 + (tid >> 4), XR, (rr) * 4)
#define X_FOLD8(c, XR, XO) X_FOLDG(c, tid >> 3, XR, XO)
#define X_FOLDG(c, MS, XR, XO)                                                                                  \
    {                                                                                                           \
        _Pragma("clang fp contract(off)")   /* same roundings in the f32 and int16 instantiations */            \
        const int ms = (MS);                                                                                    \
        const f32x4 xA = decode(XR[(XO) + 0]), xB = decode(XR[(XO) + 1]);                                       \
        const f32x4 xC = decode(XR[(XO) + 2]), xD = decode(XR[(XO) + 3]);                                       \
        const float mBx = mirror(xB.x), mDx = mirror(xD.x);                                                     \
        const f32x4 y1 = pk::mul(xA, W1), y3 = pk::mul(xC, W3);                                                 \
        const f32x4 y2 = pk::mul(f32x4{shr1(xC.x, mBx), mirror(xB.w), mirror(xB.z), mirror(xB.y)}, W3);         \
        const f32x4 y4 = pk::mul(f32x4{shr1(0.f, mDx), mirror(xD.w), mirror(xD.z), mirror(xD.y)}, W1);          \
        const f32x4 s14 = pk::add(y1, y4), d14 = pk::sub(y1, y4), s23 = pk::add(y2, y3), d23 = pk::sub(y2, y3); \
        f32x4 pe = pk::add(s14, s23), po = pk::sub(s14, s23), qe = pk::sub(d14, d23), qo = pk::add(d14, d23);   \
        {                             /* n = 0 is not part of the folded sums; rank-1 terms of n = 0 / 64 / 128 (branch-free): */ \
            /* lane q = 0 holds sample 64 (B.x), 128 (C.x) and 192 (D.x) of the column */                       \
            pe.x = q0 ? 0.f : pe.x; po.x = q0 ? 0.f : po.x; qe.x = q0 ? 0.f : qe.x; qo.x = q0 ? 0.f : qo.x;        \
            const float y64 = xB.x * w64, y192 = xD.x * w64;                                                    \
            const int fo = q0 ? (c) * 96 + ms : FCOR_SINK + lane;                                               \
            fcor[fo] = y3.x;                        /* y[128] */                                                 \
            fcor[fo + (q0 ? 32 : 0)] = y64 + y192;  /* a64 */                                                    \
            fcor[fo + (q0 ? 64 : 0)] = y64 - y192;  /* b64 */                                                    \
        }                                                                                                       \
        /* stored as two 8-byte halves each: the sums come out of the packed adds as register PAIRS, a 16-byte store      \
           would first copy them into four consecutive registers */                                            \
        if constexpr (K8) {                                                                                     \
            st2(&RX[(CS * (c) + q) * QS + ms], pe);                                                             \
            st2(&RX[(CS * (c) + QL + q) * QS + ms], po);                                                        \
            st2(&RX[(CS * (c) + 2 * QL + q) * QS + ms], qe);                                                    \
            st2(&RX[(CS * (c) + 3 * QL + q) * QS + ms], qo);                                                    \
        } else {                                                                                                \
            /* 16 kHz: the odd bins contract po | qo as they are; the even bins' operands fold once more, about n = 32      \
               (vad_layout.h, bin_of_channel_fold3): partner of n = 4q + i is 64 - n = lane 16 - q component 0 (i = 0: row_mirror \
               then row_shr:1) or lane 15 - q components 3, 2, 1 (row_mirror).  Lanes q < 8 hold n = 0..31 and store; lanes      \
               q >= 8 hold the same values again and drop them into sink rows (a select on the address, no branch: the fold   \
               stays in the MFMAs' basic block).  Slot n = 0 carries the unpaired n = 32 (lane 8, component 0): pe[32] | qe[32] */ \
            const f32x4 pm = f32x4{shr1(0.f, mirror(pe.x)), mirror(pe.w), mirror(pe.z), mirror(pe.y)};           \
            const f32x4 qm = f32x4{shr1(0.f, mirror(qe.x)), mirror(qe.w), mirror(qe.z), mirror(qe.y)};           \
            f32x4 pep = pk::add(pe, pm), pen = pk::sub(pe, pm), qen = pk::sub(qe, qm), qep = pk::add(qe, qm);   \
            const float pe32 = shl8(pe.x), qe32 = shl8(qe.x);                                                   \
            pep.x = q0 ? pe32 : pep.x; pen.x = q0 ? 0.f : pen.x; qen.x = q0 ? 0.f : qen.x; qep.x = q0 ? qe32 : qep.x; \
            st2(&RX[(CS * (c) + q) * QS + ms], po);                                                             \
            st2(&RX[(CS * (c) + 16 + q) * QS + ms], qo);                                                        \
            const int er = (q < 8 ? CS * (c) + 32 + q : ROW_FOLD_SINK - 8 + q) * QS + ms;                       \
            st2(&RX[er], pep);                                                                                  \
            st2(&RX[er + 8 * QS], pen);                                                                         \
            st2(&RX[er + 16 * QS], qen);                                                                        \
            st2(&RX[er + 24 * QS], qep);                                                                        \
        }                                                                                                       \
    }
#define H_MMA(WS, g)                                                                                            \
    {                                                                                                           \
        const f32x4 av0 = RH[(4 * (g)) * QS + hq], av1 = RH[(4 * (g) + 2) * QS + hq];                           \
        gi = mfma4(WS[0], av0, gi); gfo = mfma4(WS[1], av0, gfo); gg = mfma4(WS[2], av0, gg); go = mfma4(WS[3], av0, go); \
        gi = mfma4(WS[4], av1, gi); gfo = mfma4(WS[5], av1, gfo); gg = mfma4(WS[6], av1, gg); go = mfma4(WS[7], av1, go); \
    }
            // program order of the group that ends here: 4 x { 8 MFMA, up to 40 VALU }.  The fold's VALU work does not hide
            // under the MFMAs (same pipe) - the interleave only lets its LDS writes and the next requests overlap them -
            // and every MFMA <-> VALU alternation costs ~14 cycles (tools/ubench/mfma_valu.hip), so the pieces are coarse:
            // measured 50.0 us vs 50.4 (1 : 6 interleave) vs 50.2 (no interleave)
#define H_MIX(NV)                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, 40, 0);                                                     \
    }
            {   // the accumulators start at the gate biases: register 4 g + i of a lane = unit 8 g + 4 h + i (every lane of a half
                // wave reads the same 16 bytes: a broadcast).  The wave reads what the wave itself wrote: ahead of the barrier.
                const f32x4 *const bq = biasL + 32 * w + h;
                gi = acc_of(bq[0], bq[2], bq[4], bq[6]);
                gfo = acc_of(bq[8], bq[10], bq[12], bq[14]);
                gg = acc_of(bq[16], bq[18], bq[20], bq[22]);
                go = acc_of(bq[24], bq[26], bq[28], bq[30]);
            }
            // (0) h_{t-1} visible (wA, wB and the first column were requested in the prologue / at the end of the previous
            // frame).  For t > 0 it follows barrier (8) and costs nothing.
            __syncthreads();
            STAMP(31);
            if constexpr (!K8) {
            H_MMA(wA, 0) SB(); STAMP(20);
            H_LDW(wA, 2, wh) X_ISSUE(1, xb_, t) SB(); H_MMA(wB, 1) SB(); STAMP(21);
            H_LDW(wB, 3, wh) SB(); H_MMA(wA, 2) X_FOLD1(0, 0, xa_) H_MIX(6) SB(); STAMP(22);
            H_LDW(wA, 4, wh) SB(); H_MMA(wB, 3) X_FOLD1(0, 1, xa_) H_MIX(6) SB(); STAMP(23);
            H_LDW(wB, 5, wh) X_ISSUE(2, xa_, t) SB(); H_MMA(wA, 4) X_FOLD1(1, 0, xb_) H_MIX(6) SB(); STAMP(24);
            H_LDW(wA, 6, wh) SB(); H_MMA(wB, 5) X_FOLD1(1, 1, xb_) H_MIX(6) SB(); STAMP(25);
            H_LDW(wB, 7, wh) SB(); H_MMA(wA, 6) X_FOLD1(2, 0, xa_) H_MIX(6) SB(); STAMP(26);
            H_MMA(wB, 7) X_FOLD1(2, 1, xa_) H_MIX(6) SB(); STAMP(27);
            } else {           // 8 kHz: three columns of 32 quads, one fold call each (8 lanes per stream)
            H_MMA(wA, 0) SB();
            H_LDW(wA, 2, wh) X_ISSUE8(2, xb_, 0, t) SB(); H_MMA(wB, 1) SB();
            H_LDW(wB, 3, wh) SB(); H_MMA(wA, 2) X_FOLD8(0, xa_, 0) H_MIX(6) SB();
            H_LDW(wA, 4, wh) SB(); H_MMA(wB, 3) X_FOLD8(1, xa_, 4) H_MIX(6) SB();
            H_LDW(wB, 5, wh) SB(); H_MMA(wA, 4) X_FOLD8(2, xb_, 0) H_MIX(6) SB();
            H_LDW(wA, 6, wh) SB(); H_MMA(wB, 5) SB();
            H_LDW(wB, 7, wh) SB(); H_MMA(wA, 6) SB();
            H_MMA(wB, 7) SB();
            }
#undef H_MIX
#undef H_MMA
#undef X_FOLD1
#undef X_FOLD8
#undef X_FOLDG
        }
        // weights of the first STFT iteration are requested before the barrier (they never depend on LDS)
        f32x4 Are = WL(ws_stft), Aim = WL(ws_stft + 1);
        SB();
        STAMP(1);
        __syncthreads();   // (1) folded x and h visible
        STAMP(2);

        // ---- bin 128 (even): re = sum_n pe[n] (-1)^n + y128 + a64, im == 0; 2 lanes per (column, stream)
        {
            const int pr = lane >> 1, half = lane & 1;
            const int pair = w * 24 + pr;                 // 96 (column, stream) pairs, 24 per wave
            const int c = pair >> 5, ms = pair & 31;
            float a = 0.f;
            if (pr < 24) {
                // 16 kHz: sum_n pe[n] (-1)^n over n = 1..63 = the same alternating sum over pe[n] + pe[64 - n], n = 1..31, plus
                // pe[32] - which is what the pe+ rows hold (slot 0 = pe[32], sign +)
                constexpr int NQ = K8 ? QL / 2 : 4, R0 = K8 ? 0 : 32;
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const f32x4 pp = RX[(CS * c + R0 + half * NQ + i) * QS + ms];
                    a += (pp.x - pp.y) + (pp.z - pp.w);
                }
            }
            a += __shfl_xor(a, 1);
            if (pr < 24 && half == 0) nyqv[c * 32 + ms] = fabsf(a + fcor[(c * 3 + 0) * 32 + ms] + fcor[(c * 3 + 1) * 32 + ms]);
        }

        // ---- STFT: wave w owns the 32 bins bin_of_channel(32w + r): cos on pe|po, -sin on qe|qo, 3 columns ----
        // enc0's bias and first weights ride along (requested at the end of this phase)
        f32x4 e0b0, e0b1, e0b2, e0b3, E0w[5];
        if constexpr (K8) {
            // 8 kHz sub-model: 64 complex bins.  As two 32-row tiles they kept waves 0 / 1 busy for 96 MFMAs while waves 2 / 3 had
            // none; as FOUR 16-row tiles on v_mfma_f32_16x16x4_f32 (lane = (stream n16 + 16 sh, channel group kq), the 16-stream
            // kernel's fragment convention; pack_dft4_wave_128_t16) every wave has 96 MFMAs of half the cycles: wave w owns the
            // bins 2 (16 (w & 1) + r) + (w >> 1), r = 0..15 - waves 0 / 1 the even bins (pe / qe), 2 / 3 the odd ones (po / qo).
            // Same rank-1 start as below: tile row r = 4 kq + i, (-1)^r = (-1)^i.
            const int n16 = lane & 15, kq = lane >> 4;
            const bool even = w < 2;
            f32x4 sre[3][2], sim[3][2];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const int ms = n16 + 16 * sh;
                    const float y128 = fcor[(c * 3 + 0) * 32 + ms], a64 = fcor[(c * 3 + 1) * 32 + ms], b64 = fcor[(c * 3 + 2) * 32 + ms];
                    const float rp = even ? y128 + a64 : -y128, rm = even ? y128 - a64 : -y128;
                    const float ip = even ? 0.f : -b64, im_ = even ? 0.f : b64;
                    sre[c][sh] = f32x4{rp, rm, rp, rm};
                    sim[c][sh] = f32x4{ip, im_, ip, im_};
                }
            const int rR = even ? 0 : QL, rI = even ? 2 * QL : 3 * QL;
            const f32x4 wre0 = WL(ws_stft), wim0 = WL(ws_stft + 1), wre1 = WL(ws_stft + 2), wim1 = WL(ws_stft + 3);
            auto mma16 = [](f32x4 wv, f32x4 a, f32x4 acc) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, a.w, acc, 0, 0, 0);
                return acc;
            };
#pragma unroll
            for (int j = 0; j < 2; ++j) {                       // K = 32 folded samples = two k-iterations of 16
                const f32x4 wr = j ? wre1 : wre0, wi = j ? wim1 : wim0;
                f32x4 u[3][2], v[3][2];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int sh = 0; sh < 2; ++sh) {
                        u[c][sh] = RX[(CS * c + rR + 4 * j + kq) * QS + n16 + 16 * sh];
                        v[c][sh] = RX[(CS * c + rI + 4 * j + kq) * QS + n16 + 16 * sh];
                    }
                SB();
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int sh = 0; sh < 2; ++sh) {
                        sre[c][sh] = mma16(wr, u[c][sh], sre[c][sh]);
                        sim[c][sh] = mma16(wi, v[c][sh], sim[c][sh]);
                    }
                SB();
            }
            e0b0 = WL(ws_e0); e0b1 = WL(ws_e0 + 1); e0b2 = WL(ws_e0 + 2); e0b3 = WL(ws_e0 + 3);
#pragma unroll
            for (int p = 0; p < 5; ++p) E0w[p] = WL(ws_e0 + 4 + p);
            SB();
            STAMP(16);
            __syncthreads();   // (1b) every wave is done reading u/v: the magnitudes may overwrite them
            // magnitudes -> Toom-3 evaluations, rows 16 p + (channel / 4) = 16 p + 4 w + kq
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                f32x4 mg[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 r = sre[c][sh], i = sim[c][sh];
                    mg[c] = f32x4{mag_(r.x, i.x), mag_(r.y, i.y), mag_(r.z, i.z), mag_(r.w, i.w)};
                }
                const f32x4 s02 = mg[0] + mg[2];
                f32x4 *o = RX + (4 * w + kq) * QS + n16 + 16 * sh;
                st2(o, mg[0]);
                st2(o + PS * QS, s02 + mg[1]);
                st2(o + 2 * PS * QS, s02 - mg[1]);
                st2(o + 3 * PS * QS, f32x4{fmaf(4.f, mg[2].x, fmaf(2.f, mg[1].x, mg[0].x)), fmaf(4.f, mg[2].y, fmaf(2.f, mg[1].y, mg[0].y)),
                                           fmaf(4.f, mg[2].z, fmaf(2.f, mg[1].z, mg[0].z)), fmaf(4.f, mg[2].w, fmaf(2.f, mg[1].w, mg[0].w))});
                st2(o + 4 * PS * QS, mg[2]);
            }
            if (tid < 64) {    // |X64|: as |X128| below
                const float n0 = nyqv[m], n1 = nyqv[32 + m], n2 = nyqv[64 + m];
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                RX[ROWN * QS + hq] = h == 0 ? f32x4{n0, (n0 + n2) + n1, (n0 + n2) - n1, fmaf(4.f, n2, fmaf(2.f, n1, n0))} : z4;
                RX[(ROWN + 2) * QS + hq] = h == 0 ? f32x4{n2, 0.f, 0.f, 0.f} : z4;
            }
        } else
        {
            // 16 kHz: wave w owns the channels 32 w .. 32 w + 31 as TWO 16-row tiles on v_mfma_f32_16x16x4_f32 (lane = (stream
            // n16 + 16 sh, channel group kq); vad_layout.h, bin_of_channel_fold3): row tile 0 = 16 odd bins, cos on po, -sin on qo,
            // K = 64 (k-iterations 0..3); row tile 1 = 16 even bins on the once-more-folded operands pe+- | qe-+ (waves 2, 3 | waves
            // 0, 1), K = 32 (k-iterations 4, 5).  288 half-length MFMAs per wave instead of 192 full ones: three quarters.
            // The accumulators START from the rank-1 terms of n = 0, 64, 128 (the samples the fold cannot pair), so the epilogue
            // only takes magnitudes; tile row r = 4 kq + i, (-1)^r = (-1)^i:
            //   odd k: re = -y128, im = -+ b64 (sin(pi k / 2) alternates along the rows); even k = 2 m: re = y128 + (-1)^m a64, im = 0
            const int n16 = lane & 15, kq = lane >> 4;
            const bool mo = w < 2;                           // this wave's even tile: m odd | m even
            f32x4 sre[3][2][2], sim[3][2][2];                // [column][stream half][row tile]
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const int ms = n16 + 16 * sh;
                    const float y128 = fcor[(c * 3 + 0) * 32 + ms], a64 = fcor[(c * 3 + 1) * 32 + ms], b64 = fcor[(c * 3 + 2) * 32 + ms];
                    const float re1 = mo ? y128 - a64 : y128 + a64;
                    sre[c][sh][0] = f32x4{-y128, -y128, -y128, -y128};
                    sim[c][sh][0] = f32x4{-b64, b64, -b64, b64};
                    sre[c][sh][1] = f32x4{re1, re1, re1, re1};
                    sim[c][sh][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            auto mma16 = [](f32x4 wv, f32x4 a, f32x4 acc) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, a.w, acc, 0, 0, 0);
                return acc;
            };
            const int eR = mo ? 40 : 32, eI = mo ? 56 : 48;
            const f32x4 *const XB = RX + kq * QS + n16;
            // operand rows of k-iteration t: odd tile t = 0..3: po rows 4 t + kq, qo rows 16 + 4 t + kq; even tile t = 4, 5
#define S_ROW_R(t) ((t) < 4 ? 4 * (t) : eR + 4 * ((t) - 4))
#define S_ROW_I(t) ((t) < 4 ? 16 + 4 * (t) : eI + 4 * ((t) - 4))
            f32x4 Awr = Are, Awi = Aim, Bwr, Bwi;            // Are / Aim: the first k-iteration's blocks, requested before barrier (1)
            f32x4 Au[3][2], Av[3][2], Bu[3][2], Bv[3][2];
#define S_LDX(S, tt)                                                                       \
    _Pragma("unroll") for (int c = 0; c < 3; ++c)                                          \
        _Pragma("unroll") for (int sh = 0; sh < 2; ++sh) {                                 \
            S##u[c][sh] = XB[(CS * c + S_ROW_R(tt)) * QS + 16 * sh];                       \
            S##v[c][sh] = XB[(CS * c + S_ROW_I(tt)) * QS + 16 * sh];                       \
        }
#define S_LD(S, tt) S##wr = WL(ws_stft + 2 * (tt)); S##wi = WL(ws_stft + 2 * (tt) + 1); S_LDX(S, tt)
    // (program order: the twelve accumulators in turn for each of a quad's four components - consecutive MFMAs are independent)
#define S_MMA(S, rt)                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
        _Pragma("unroll") for (int c = 0; c < 3; ++c)                                      \
            _Pragma("unroll") for (int sh = 0; sh < 2; ++sh) {                             \
                sre[c][sh][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(S##wr[j], S##u[c][sh][j], sre[c][sh][rt], 0, 0, 0); \
                sim[c][sh][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(S##wi[j], S##v[c][sh][j], sim[c][sh][rt], 0, 0, 0); \
            }
            // a k-iteration's 48 MFMAs (32 cycles each) with the NEXT iteration's requests - 2 weight blocks, 12 LDS reads - issued in
            // their shadow, one LDS read behind every four MFMAs: issued as a burst between the groups (as the other phases, whose
            // MFMAs are twice as long and whose requests are half as many, still do) they cost ~430 cycles per group, 2.6 k of this
            // phase's 11.8 k
#define S_IL                                                                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
    }
            S_LDX(A, 0)
            SB();
            S_LD(B, 1) S_MMA(A, 0) S_IL SB();
            S_LD(A, 2) S_MMA(B, 0) S_IL SB();
            S_LD(B, 3) S_MMA(A, 0) S_IL SB();
            S_LD(A, 4) S_MMA(B, 0) S_IL SB();
            S_LD(B, 5) S_MMA(A, 1) S_IL SB();
            S_MMA(B, 1) SB();
#undef S_IL
#undef S_LDX
#undef S_LD
#undef S_MMA
#undef S_ROW_R
#undef S_ROW_I
            // request enc0's bias + first weights now: they land while the magnitudes are written
            e0b0 = WL(ws_e0); e0b1 = WL(ws_e0 + 1); e0b2 = WL(ws_e0 + 2); e0b3 = WL(ws_e0 + 3);
#pragma unroll
            for (int p = 0; p < 5; ++p) E0w[p] = WL(ws_e0 + 4 + p);
            SB();
            STAMP(16);
            __syncthreads();   // (1b) every wave is done reading u/v: the magnitudes may overwrite them
            // magnitudes.  The three columns m0, m1, m2 of a bin go to enc0 as the Toom-3 evaluations of m0 + m1 z + m2 z^2
            // (vad_layout.h): rows 32 p + channel / 4 = 32 p + 8 w + 4 rt + kq, p = 0..4 for z = 0, 1, -1, 2, inf
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    f32x4 mg[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) mg[c] = pk::mag(sre[c][sh][rt], sim[c][sh][rt]);
                    const f32x4 s02 = pk::add(mg[0], mg[2]);
                    f32x4 *o = RX + (8 * w + 4 * rt + kq) * QS + n16 + 16 * sh;
                    st2(o, mg[0]);
                    st2(o + PS * QS, pk::add(s02, mg[1]));
                    st2(o + 2 * PS * QS, pk::sub(s02, mg[1]));
                    st2(o + 3 * PS * QS, pk::fma(pk::splat(4.f), mg[2], pk::fma(pk::splat(2.f), mg[1], mg[0])));
                    st2(o + 4 * PS * QS, mg[2]);
                }
            // |X128|: rows 160 / 161 = (points 0, 1, -1, 2) / zeros, rows 162 / 163 = (inf, 0, 0, 0) / zeros
            if (tid < 64) {
                const float n0 = nyqv[m], n1 = nyqv[32 + m], n2 = nyqv[64 + m];
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                RX[ROWN * QS + hq] = h == 0 ? f32x4{n0, (n0 + n2) + n1, (n0 + n2) - n1, fmaf(4.f, n2, fmaf(2.f, n1, n0))} : z4;
                RX[(ROWN + 2) * QS + hq] = h == 0 ? f32x4{n2, 0.f, 0.f, 0.f} : z4;
            }
        }
        STAMP(3);
        __syncthreads();   // (2) magnitudes complete
        STAMP(4);

        // ---- enc0: 129 -> 128 ch, k3 s1 p1, 3 -> 3 columns; wave w: channels 32w..  Toom-3: five point-wise
        //      contractions over the 129 channels instead of seven (tap, column) ones, then the interpolation ---------
        f32x4 e1b0, e1b1, e1b2, e1b3, E1w0, E1w1, E1w2, E1w3;
        {
            const int ws = ws_e0 + 4;
            f32x16 acc[5];
#pragma unroll
            for (int p = 0; p < 5; ++p) acc[p] = (f32x16)(0.f);
            f32x4 Aw[5], Bw[5], Aa[5], Ba[5];
#pragma unroll
            for (int p = 0; p < 5; ++p) { Aw[p] = E0w[p]; Aa[p] = RX[(PS * p) * QS + hq]; }
#define E0_LD(S, jj)                                                                       \
    _Pragma("unroll") for (int p = 0; p < 5; ++p) {                                        \
        S##w[p] = WL(ws + 5 * (jj) + p);                                                   \
        S##a[p] = RX[(PS * p + 2 * (jj)) * QS + hq];                                       \
    }
#define E0_MMA(S) _Pragma("unroll") for (int p = 0; p < 5; ++p) acc[p] = mfma4(S##w[p], S##a[p], acc[p]);
            for (int j = 0; j < NJ0; j += 2) {
                E0_LD(B, j + 1) SB();
                E0_MMA(A) SB();
                const int jn = j + 2 < NJ0 ? j + 2 : NJ0 - 2;
                E0_LD(A, jn) SB();
                E0_MMA(B) SB();
            }
#undef E0_LD
#undef E0_MMA
            STAMP(17);
            {   // input channel 128 (Nyquist bin): one MFMA per point (K = 2 with the upper half-wave at zero)
                const f32x4 an = RX[ROWN * QS + hq], bn = RX[(ROWN + 2) * QS + hq];
                const f32x4 wna = WL(ws + 5 * NJ0), wnb = WL(ws + 5 * NJ0 + 1);
                // enc1's bias and first group of weights
                e1b0 = WL(ws_e1); e1b1 = WL(ws_e1 + 1); e1b2 = WL(ws_e1 + 2); e1b3 = WL(ws_e1 + 3);
                E1w0 = WL(ws_e1 + 4); E1w1 = WL(ws_e1 + 5); E1w2 = WL(ws_e1 + 6); E1w3 = WL(ws_e1 + 7);
                SB();
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wna.x, an.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wna.y, an.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wna.z, an.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wna.w, an.w, acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wnb.x, bn.x, acc[4], 0, 0, 0);
            }
            // interpolation (P(1), P(-1) arrive halved: the 1/2 sits in the weights) + bias
            {
                // (packed: sixteen registers per tile and step - written with the scalar operators every subtraction here became
                // sixteen v_sub_f32)
                const f32x16 bias = acc_of(e0b0, e0b1, e0b2, e0b3);
                const f32x16 y0 = acc[0], y4 = acc[4];
                const f32x16 bb = pk::sub16(acc[1], acc[2]);
                const f32x16 y2 = pk::sub16(pk::sub16(pk::add16(acc[1], acc[2]), y0), y4);
                const f32x16 t2 = pk::fma16(y4, -16.0f, pk::fma16(y2, -4.0f, pk::sub16(acc[3], y0)));          // = 2 (y1 + 4 y3)
                const f32x16 y3 = pk::fma16(bb, -1.0f / 3.0f, pk::mul16(t2, 1.0f / 6.0f));
                acc[0] = pk::add16(pk::sub16(bb, y3), bias);
                acc[1] = pk::add16(y2, bias);
                acc[2] = pk::add16(y3, bias);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) store_tile_relu(RE, c * 32 + 8 * w, m, h, acc[c]);
        }
        STAMP(5);
        __syncthreads();   // (3) enc0 out in A; region B free
        STAMP(6);

        // thin layers: one accumulator chain per wave; a "group" = 4 k-iterations (4 weight blocks,
        // 4 activation quads, 16 MFMAs), ping-pong between two register sets
#define G_LDW(S, ws_, g4)                                                                  \
    S##w0 = WL((ws_) + (4 * (g4))); S##w1 = WL((ws_) + (4 * (g4) + 1));            \
    S##w2 = WL((ws_) + (4 * (g4) + 2)); S##w3 = WL((ws_) + (4 * (g4) + 3));
#define G_MMA(S) acc = mfma4(S##w0, S##a0, acc); acc = mfma4(S##w1, S##a1, acc); acc = mfma4(S##w2, S##a2, acc); acc = mfma4(S##w3, S##a3, acc);

        // ---- enc1: 128 -> 64 ch, k3 s2 p1, 3 -> 2 columns; wave w: n-tile w&1, column w>>1 ----
        f32x4 e2b0, e2b1, e2b2, e2b3, E2w0, E2w1, E2w2, E2w3;
        f32x4 e3b0, e3b1, e3b2, e3b3, E3w0, E3w1, E3w2, E3w3;
        {
            const int nt = w & 1, tp = w >> 1;
            const int ws = ws_e1 + 4;
            f32x16 acc = acc_of(e1b0, e1b1, e1b2, e1b3);
            // valid taps: tp=0 -> taps 1,2 on input columns 0,1 ; tp=1 -> taps 0,1 on columns 1,2
            // iteration it in 0..31: column tp + (it>>4), quad pair it&15
#define E1_ROW(it) ((tp + ((it) >> 4)) * 32 + 2 * ((it) & 15))
#define E1_LDA(S, g4)                                                                      \
    S##a0 = RE[E1_ROW(4 * (g4)) * QS + hq]; S##a1 = RE[E1_ROW(4 * (g4) + 1) * QS + hq];    \
    S##a2 = RE[E1_ROW(4 * (g4) + 2) * QS + hq]; S##a3 = RE[E1_ROW(4 * (g4) + 3) * QS + hq];
            f32x4 Aw0 = E1w0, Aw1 = E1w1, Aw2 = E1w2, Aw3 = E1w3, Bw0, Bw1, Bw2, Bw3;
            f32x4 Aa0, Aa1, Aa2, Aa3, Ba0, Ba1, Ba2, Ba3;
            E1_LDA(A, 0)
            for (int g4 = 0; g4 < 8; g4 += 2) {
                G_LDW(B, ws, g4 + 1) E1_LDA(B, g4 + 1) SB();
                G_MMA(A) SB();
                const int gn = g4 + 2 < 8 ? g4 + 2 : 6;
                G_LDW(A, ws, gn) E1_LDA(A, gn) SB();
                if (g4 == 6) {   // last pass: request enc2's (bias: K half 0 only) and enc3's first blocks
                    const int ge = 4 + 8 * (w >> 1);           // this wave's K half: weight groups 2 (w>>1), 2 (w>>1) + 1
                    e2b0 = WL(ws_e2); e2b1 = WL(ws_e2 + 1); e2b2 = WL(ws_e2 + 2); e2b3 = WL(ws_e2 + 3);
                    E2w0 = WL(ws_e2 + ge); E2w1 = WL(ws_e2 + ge + 1); E2w2 = WL(ws_e2 + ge + 2); E2w3 = WL(ws_e2 + ge + 3);
                    e3b0 = WL(ws_e3); e3b1 = WL(ws_e3 + 1); e3b2 = WL(ws_e3 + 2); e3b3 = WL(ws_e3 + 3);
                    E3w0 = WL(ws_e3 + 4); E3w1 = WL(ws_e3 + 5); E3w2 = WL(ws_e3 + 6); E3w3 = WL(ws_e3 + 7);
                    SB();
                }
                G_MMA(B) SB();
            }
#undef E1_ROW
#undef E1_LDA
            store_tile_relu(RX, tp * 16 + 8 * nt, m, h, acc);
        }
        STAMP(7);
        __syncthreads();   // (4) enc1 out in B; region A free
        STAMP(8);

        // ---- enc2: 64 -> 64 ch, k3 s2 p1, 2 -> 1 column (taps 1,2 on columns 0,1) ----
        // split-K over the 4 waves: wave w = output tile w&1, K half w>>1 (= input column w>>1).  The two partial
        // tiles (bias in half 0) go to rows 16 (w>>1) + 8 (w&1) un-activated; enc3 adds them and applies the ReLU
        // as it reads its input - no extra barrier.
        {
            const int kh2 = w >> 1;
            const int ws = ws_e2 + 4;
            f32x16 acc = kh2 == 0 ? acc_of(e2b0, e2b1, e2b2, e2b3) : (f32x16)(0.f);
            // iteration it in 0..15: column it>>3, quad pair it&7
#define E2_ROW(it) (((it) >> 3) * 16 + 2 * ((it) & 7))
#define E2_LDA(S, g4)                                                                      \
    S##a0 = RX[E2_ROW(4 * (g4)) * QS + hq]; S##a1 = RX[E2_ROW(4 * (g4) + 1) * QS + hq];    \
    S##a2 = RX[E2_ROW(4 * (g4) + 2) * QS + hq]; S##a3 = RX[E2_ROW(4 * (g4) + 3) * QS + hq];
            f32x4 Aw0 = E2w0, Aw1 = E2w1, Aw2 = E2w2, Aw3 = E2w3, Bw0, Bw1, Bw2, Bw3;
            f32x4 Aa0, Aa1, Aa2, Aa3, Ba0, Ba1, Ba2, Ba3;
            E2_LDA(A, 2 * kh2)
            G_LDW(B, ws, 2 * kh2 + 1) E2_LDA(B, 2 * kh2 + 1) SB();
            G_MMA(A) SB();
            G_MMA(B) SB();
#undef E2_ROW
#undef E2_LDA
#pragma unroll
            for (int g = 0; g < 4; ++g) RE[(16 * kh2 + 8 * (w & 1) + 2 * g) * QS + hq] = quad_of(acc, g);
        }
        STAMP(9);
        __syncthreads();   // (5) enc2 out in A
        STAMP(10);

        // ---- enc3: 64 -> 128 ch, k3 s1 p1 on a single column: centre tap only ------------------
        f32x4 Lw0, Lw1, Lw2, Lw3;
        {
            const int ws = ws_e3 + 4;
            f32x16 acc = acc_of(e3b0, e3b1, e3b2, e3b3);
            f32x4 Aw0 = E3w0, Aw1 = E3w1, Aw2 = E3w2, Aw3 = E3w3, Bw0, Bw1, Bw2, Bw3;
            // input quad r = relu(enc2 partial of K half 0 (row r) + K half 1 (row 16 + r))
            auto e2q = [&](int r) -> f32x4 {
                const f32x4 a = RE[r * QS + hq], b2 = RE[(16 + r) * QS + hq];
                return relu4(f32x4{a.x + b2.x, a.y + b2.y, a.z + b2.z, a.w + b2.w});
            };
            f32x4 Aa0 = e2q(0), Aa1 = e2q(2), Aa2 = e2q(4), Aa3 = e2q(6);
            G_LDW(B, ws, 1)
            const f32x4 Ba0 = e2q(8), Ba1 = e2q(10), Ba2 = e2q(12), Ba3 = e2q(14);
            // the first weight blocks of the LSTM's input half
            Lw0 = WL(ws_l + 16); Lw1 = WL(ws_l + 17); Lw2 = WL(ws_l + 18); Lw3 = WL(ws_l + 19);
            SB();
            G_MMA(A) SB();
            G_MMA(B) SB();
            store_tile_relu(RX, 8 * w, m, h, acc);
        }
#undef G_LDW
#undef G_MMA
        STAMP(11);
        __syncthreads();   // (6) enc3 out (LSTM input x) in B rows 0..31
        STAMP(12);

        // ---- LSTM cell: wave w owns hidden units 32w..32w+31, all four gates -------------------
        {
            const int ws = ws_l + 16;
            // the recurrent half (W_hh . h_{t-1}, + bias) was accumulated while the frame was loading;
            // iteration it in 0..15 contracts x = enc3 output (rows 0..31)
#define L_ROW(it) (RX + (2 * (it)) * QS + hq)
#define L_LD(S, it)                                                                        \
    S##wi = WL(ws + (4 * (it))); S##wf = WL(ws + (4 * (it) + 1));                  \
    S##wg = WL(ws + (4 * (it) + 2)); S##wo = WL(ws + (4 * (it) + 3)); S##av = *L_ROW(it);
#define L_MMA(S) gi = mfma4(S##wi, S##av, gi); gfo = mfma4(S##wf, S##av, gfo); gg = mfma4(S##wg, S##av, gg); go = mfma4(S##wo, S##av, go);
            f32x4 Awi = Lw0, Awf = Lw1, Awg = Lw2, Awo = Lw3, Aav = RX[hq], Bwi, Bwf, Bwg, Bwo, Bav;
            f32x4 hw0, hw1, hw2, hw3;
            for (int it = 0; it < 16; it += 2) {
                L_LD(B, it + 1) SB();
                L_MMA(A) SB();
                const int itn = it + 2 < 16 ? it + 2 : 14;
                L_LD(A, itn) SB();
                if (it == 14) {   // head weights for the epilogue
                    hw0 = WL(ws + 128); hw1 = WL(ws + 129); hw2 = WL(ws + 130); hw3 = WL(ws + 131);
                    SB();
                }
                L_MMA(B) SB();
            }
#undef L_ROW
#undef L_LD
#undef L_MMA
            STAMP(18);
            STAMP(13);
            __syncthreads();   // (7) every wave is done reading h_{t-1}; region A free for the next frame
            STAMP(14);
            f32x4 part4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 i4 = quad_of(gi, g), f4 = quad_of(gfo, g), g4 = quad_of(gg, g), o4 = quad_of(go, g);
                const f32x4 c4 = quad_of(cst, g);
                const f32x4 hw = g == 0 ? hw0 : (g == 1 ? hw1 : (g == 2 ? hw2 : hw3));
                // c' = sigma(f) c + sigma(i) tanh(g); h' = sigma(o) tanh(c'); head partial += w relu(h') - a quad at a time, the
                // full-rate arithmetic packed (pk::), the transcendentals per component
                const f32x4 cn = pk::fma(pk::sigmoid4(f4), c4, pk::mul(pk::sigmoid4(i4), pk::tanh4(g4)));
                const f32x4 hn = pk::mul(pk::sigmoid4(o4), pk::tanh4(cn));
                part4 = pk::fma(hw, relu4(hn), part4);
                RH[(8 * w + 2 * g) * QS + hq] = hn;
                if (t == T - 1 && live) {   // last frame of the call: h' and c' go back to HBM under barrier (8), head and state machine
                    *reinterpret_cast<f32x4 *>(KP(state) + (size_t)slot * 256 + 32 * w + 8 * g + 4 * h) = hn;
                    *reinterpret_cast<f32x4 *>(KP(state) + (size_t)slot * 256 + 128 + 32 * w + 8 * g + 4 * h) = cn;
                }
                switch (g) {
                    case 0: cst.s0 = cn.x; cst.s1 = cn.y; cst.s2 = cn.z; cst.s3 = cn.w; break;
                    case 1: cst.s4 = cn.x; cst.s5 = cn.y; cst.s6 = cn.z; cst.s7 = cn.w; break;
                    case 2: cst.s8 = cn.x; cst.s9 = cn.y; cst.sa = cn.z; cst.sb = cn.w; break;
                    default: cst.sc = cn.x; cst.sd = cn.y; cst.se = cn.z; cst.sf = cn.w; break;
                }
            }
            float part = (part4.x + part4.y) + (part4.z + part4.w);
            part += __shfl_xor(part, 32);
            if (h == 0) headp[w * 32 + m] = part;
        }
        __syncthreads();   // (8) head partials + new h visible
        STAMP(15);

        // ---- head: p = sigmoid(b + sum_j w_j relu(h'_j)); then the state machine ---------------
        if (tid < MT) {
            const float z = hb + ((headp[tid] + headp[32 + tid]) + (headp[64 + tid] + headp[96 + tid]));
            const float p = fminf(sigmoidf_(z), 1.0f);
            if (sm_thread) {
                P.probs[(size_t)(tile0 + tid) * T + t] = p;
                SmSlot sm = smL[tid];
                int seg = 0;
                const int ev = sm_step(sm, p, &seg);
                if (t == T - 1) KP(sm)[sm_slot] = sm;
                else smL[tid] = sm;
                if (ev & 2) seg_last = seg;
                if (P.events) P.events[(size_t)(tile0 + tid) * T + t] = (uint8_t)ev;
            }
        }
        STAMP(28);
        // no barrier needed here: the next frame's loader only writes region A (free since (7)),
        // headp is rewritten only after barriers (1)..(7) of the next frame.
        if (++t >= T) break;
        H_FIRST(o_l, t)                            // the next frame's first requests
    }

#undef H_FIRST
#undef H_LDW
#undef X_ISSUE
#undef X_ISSUE8
    // ---- epilogue ----
    if (sm_thread && P.seg_frames) P.seg_frames[tile0 + tid] = seg_last;
}

#undef KP
// host-callable launcher (engine.cpp is plain C++ and never sees <<<>>>)
extern "C" hipError_t vadk_launch_silero_v5(const vadk::StepParams *p, hipStream_t stream) {
    (void)hipGetLastError();   // HIP's last-error slot is sticky and process-wide: a stale failure from anywhere else must not become ours
    const int tiles = (p->n + vadk::MT - 1) / vadk::MT;
    if (tiles <= 0) return hipSuccess;
#define V5_ARGS p->wstream, p->state, p->sm, p->slots, p->frames, (int)p->n, p->wstream_bytes, (int)p->T, *p
#define V5_LAUNCH(F, K, O) hipLaunchKernelGGL((silero_v5_step<F, K, O>), dim3(tiles), dim3(vadk::NTHREADS), 0, stream, V5_ARGS)
    if (p->T < 1) return hipErrorInvalidValue;     // the frame loop tests its count at the bottom
    const bool k8 = p->variant == 1;               // the 8 kHz sub-model, 256-sample frames
    const bool f32 = p->fmt == 0, one = p->T == 1;
    if (k8) {
        if (f32) { if (one) V5_LAUNCH(true, true, true); else V5_LAUNCH(true, true, false); }
        else { if (one) V5_LAUNCH(false, true, true); else V5_LAUNCH(false, true, false); }
    } else {
        if (f32) { if (one) V5_LAUNCH(true, false, true); else V5_LAUNCH(true, false, false); }
        else { if (one) V5_LAUNCH(false, false, true); else V5_LAUNCH(false, false, false); }
    }
#undef V5_LAUNCH
#undef V5_ARGS
    return hipGetLastError();
}

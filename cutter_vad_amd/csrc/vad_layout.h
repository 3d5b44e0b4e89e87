// Shared host/device layout constants of the fused Silero kernels (gfx950).
//
// Geometry common to every model kernel:
//   * one workgroup = 4 wavefronts (256 threads) = one tile of MT = 32 streams;
//   * every contraction runs on v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand
//     (32 output channels on the rows) and the ACTIVATIONS as the B operand (32 streams on
//     the columns), so the D tile has the stream on the lane (col = lane & 31) and 16 output
//     channels in the lane's registers: channel = 8*(r>>2) + 4*(lane>>5) + (r&3);
//   * activations live in LDS as "quads": row q holds channels 4q..4q+3 of all 32 streams
//     as 32 float4 (+1 float4 of padding, row stride QS = 33 float4 = 528 B, which makes both
//     the transposing ds_write_b128 of the loader and the row reads conflict-free);
//   * weights are pre-packed on the host into per-wave linear streams of 1 KiB blocks
//     (64 lanes x float4) in exactly the order the wave consumes them, so a wave's weight
//     traffic is one coalesced global_load_dwordx4 per 4 MFMAs, straight into VGPRs
//     (each wave owns different output channels, so LDS staging would not be shared).
//   One k-iteration j consumes quad 2j (lanes 0-31) and quad 2j+1 (lanes 32-63): MFMA
//   k-step i in 0..3 contracts channel 8j+i (lower half-wave) and 8j+4+i (upper half).
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

namespace vadk {

constexpr int MT = 32;        // streams per workgroup
constexpr int NWAVES = 4;     // wavefronts per workgroup
constexpr int NTHREADS = 256;
constexpr int QS = 33;        // float4 per LDS quad row
constexpr int BLK_F4 = 64;    // float4 per weight block (one per lane)
constexpr int BLK_FLOATS = 256;

// ---- Silero V5 (16 kHz branch) -------------------------------------------------------
namespace v5 {
// STFT as a 4-way folded DFT.  The stored basis is the windowed DFT w[n] cos/sin(2 pi k n / 256) (the packer
// verifies this to 2e-7 and takes w[n] from its k = 0 row), so with y = w * x and, for n = 1..63,
//   pe = y[n] + y[256-n] + y[128-n] + y[128+n]      po = y[n] + y[256-n] - y[128-n] - y[128+n]
//   qe = y[n] - y[256-n] - y[128-n] + y[128+n]      qo = y[n] - y[256-n] + y[128-n] - y[128+n]
//   re[k even] =  sum pe cos + y128 + a64 (-1)^(k/2)        re[k odd] =  sum po cos - y128
//   im[k even] = -sum qe sin                                im[k odd] = -sum qo sin - b64 (-1)^((k-1)/2)
// (y128 = y[128], a64 = y[64] + y[192], b64 = y[64] - y[192]): every contraction has K = 64 instead of 256.
// Wave w owns 32 bins: w = 0: k = 0,2..62   w = 1: k = 64..126   w = 2: k = 1,3..63   w = 3: k = 65..127,
// so |STFT| channel ch = 32 w + r holds bin  (w < 2 ? 64 w + 2 r : 64 (w - 2) + 2 r + 1); enc0's input
// channels are permuted accordingly by the packer.  Bin 128 (even) is an alternating sum on the VALU.
// weight-stream sections, in blocks, per wave
constexpr int STFT_BLOCKS = 16;            // 8 k-iterations x {cos, -sin}
constexpr int NYQ_BLOCKS = 1;              // shared table: floats 0..255 = w[n] (the Hann window of the stored basis)
constexpr int ENC0_BLOCKS = 4 + 16 * 5 + 2;  // bias, 16 k-iterations x 5 Toom-3 points, Nyquist channel (points 0,1,-1,2 | inf)
constexpr int ENC1_BLOCKS = 4 + 2 * 16;
constexpr int ENC2_BLOCKS = 4 + 2 * 8;     // packed for waves 0,1; waves 2,3 read the second K half of the same streams
constexpr int ENC3_BLOCKS = 4 + 8;
constexpr int LSTM_BLOCKS = 16 + 64 + 64 + 4;  // bias(4 gates), W_ih, W_hh, head weights
enum Section { S_STFT = 0, S_NYQ, S_ENC0, S_ENC1, S_ENC2, S_ENC3, S_LSTM, S_HEADB, S_COUNT };  // S_HEADB: 1 block, float 0 = head bias
__host__ __device__ constexpr int bin_of_channel(int ch) {
    return (ch >> 5) < 2 ? 64 * (ch >> 5) + 2 * (ch & 31) : 64 * ((ch >> 5) - 2) + 2 * (ch & 31) + 1;
}
// Channel order of the kernels whose STFT folds the EVEN bins once more (silero_v5.hip's 16 kHz instantiation, silero_v4_t16.hip):
// wave w owns the channels 32 w .. 32 w + 31 as two 16-row tiles of v_mfma_f32_16x16x4_f32.  Row tile 0: the odd bins
// 2 (16 w + r) + 1 against the 4-way folded operands po | qo (K = 64).  Row tile 1: even bins k = 2 m, whose operands fold again
// about n = 32 - cos(2 pi m (64 - n) / 128) = (-1)^m cos(2 pi m n / 128), the sine with the opposite sign - into K = 32, in two
// classes by the parity of m: waves 0, 1: k = 4 (16 w + r) + 2 (m odd: pe[n] - pe[64 - n] | qe[n] + qe[64 - n]), waves 2, 3:
// k = 4 (16 (w - 2) + r) (m even: pe[n] + pe[64 - n] | qe[n] - qe[64 - n]; bin 0 = wave 2, tile 1, row 0).  Per column a wave
// contracts 16 x (64 + 64) + 16 x (32 + 32) instead of 32 x (64 + 64): three quarters of the 4-way fold's MFMAs, the same on
// every wave.
//   loader (v5, 16 kHz): column c: po -> rows 64c + q, qo -> 64c + 16 + q (q = 0..15, n = 4q..4q+3), then n = 4q'..4q'+3, q' = 0..7:
//   pe+ -> 64c + 32 + q', pe- -> 64c + 40 + q', qe- -> 64c + 48 + q', qe+ -> 64c + 56 + q'; slot n = 0 of pe+ holds pe[32], of qe+ qe[32]
__host__ __device__ constexpr int bin_of_channel_fold3(int ch) {
    return (ch & 16) == 0 ? 2 * (16 * (ch >> 5) + (ch & 15)) + 1
                          : ((ch >> 5) < 2 ? 4 * (16 * (ch >> 5) + (ch & 15)) + 2 : 4 * (16 * ((ch >> 5) - 2) + (ch & 15)));
}
constexpr int ROW_FOLD_SINK = 192;         // 32 rows past the three columns: where the loader lanes q >= 8 drop their duplicate quads
// The graph's 8 kHz sub-model (If_0 else-branch) is the same dataflow at half the front-end size: 256-sample
// frames, window N = 128, hop 64 -> three columns at 0, 64, 128; 65 bins; 4-way fold with n = 1..31 (K = 32), unpaired samples
// n = 0, 32, 64; the 64 complex bins are two 32-row tiles - wave 0: even bins 2 r, wave 1: odd bins 2 r + 1 - and bin 64
// (even) is the alternating sum on the VALU; encoder.0 has 65 input channels.
//   loader : column c: pe -> rows 32c + q, po -> 32c + 8 + q, qe -> 32c + 16 + q, qo -> 32c + 24 + q   (q = 0..7, n = 4q..4q+3)
//   |STFT| : Toom-3 planes rows 16p + ch/4 (p = 0..4), rows 80/81 and 82/83 for |X64| as ROW_NYQ below
__host__ __device__ constexpr int bin_of_channel_8k(int ch) { return 2 * (ch & 31) + (ch >> 5); }
constexpr int ROW_NYQ_8K = 80;
constexpr int FRAME_8K = 256;

// LDS, in quad rows.  One activation region X, reused by every layer:
//   loader : column c: pe -> rows 64c + q, po -> 64c + 16 + q, qe -> 64c + 32 + q, qo -> 64c + 48 + q   (q = 0..15, n = 4q..4q+3)
//   |STFT| : the three columns x0, x1, x2 of a channel enter enc0 as the Toom-3 evaluations of x0 + x1 z + x2 z^2 at
//            z = 0, 1, -1, 2, inf: rows 32p + ch/4 (p = 0..4); rows 160/161 = (the same for |X128|, points 0,1,-1,2) / 0,
//            rows 162/163 = (|X128| at inf, 0, 0, 0) / 0
//   enc0   : rows 164 + 32c + ch/4         enc1 : rows 16c + ch/4
//   enc2   : rows 164 + ch/4 (two K halves) enc3 : rows ch/4 (LSTM input)
// enc0 as a Toom-3 product: out(c) = sum_tap w[tap] x[c + tap - 1] are the coefficients y1, y2, y3 of
// (w2 + w1 z + w0 z^2)(x0 + x1 z + x2 z^2); five point-wise products (one MFMA contraction over the channels each)
// replace the seven (tap, column) contractions: y0 = P(0), y4 = P(inf), y2 = (P(1) + P(-1))/2 - y0 - y4,
// b = (P(1) - P(-1))/2, y3 = ((P(2) - y0 - 4 y2 - 16 y4)/2 - b)/3, y1 = b - y3.  fp32 error 1.5-2.5 x that of the direct
// sums on real spectra (tools note in DESIGN.md), 28 % fewer MFMAs in the kernel's largest phase.
constexpr int ROWS_X = 260;
constexpr int ROW_E = 164;                 // first row of the upper part (enc0 / enc2 outputs)
constexpr int ROW_NYQ = 160;
constexpr int ROWS_H = 32;                 // h_{t-1}
constexpr int LDS_F4 = (ROWS_X + ROWS_H) * QS + 32 + 24 + 72 + 16 + 192 + 128;  // + head partials [4][32], |X128| [3][32], fold corrections [3][3][32], write sink [64], state machines [32] x 96 B, gate biases [4 waves][4 gates][32 units]
constexpr int LSTM_BIAS_BLOCK = 16 + 64 + 64 + 4;   // block of a wave's LSTM section that holds its gate biases compact: floats [gate][unit]
constexpr int LDS_BYTES = LDS_F4 * 16;
}  // namespace v5

// ---- Silero V4 (16 kHz branch) -------------------------------------------------------
// One launch per frame, two LDS layouts in sequence (the 258 x 8 first-layer input of 32 streams does not fit one CU's
// LDS next to the STFT operands, so the magnitudes wait in registers until those are dead):
//   STFT part : gate/int16 ingest, reflect pad 96+96, fold, 8-column STFT (MFMA), |.| -> registers
//   tail      : log-spectrum + adaptive normalisation, 4 separable blocks, 2 x LSTM(64), head, state machine
namespace v4 {
enum Section {
    S_STFT = 0, S_NYQ,            // as V5 (the DFT basis is identical): folded-DFT tables, window table
    S_DW0,                        // first-layer depthwise taps+bias per channel quad (VALU table)
    S_L0,                         // first layer: pw(relu(dw(x1))) + proj(x1), 16 outputs (rows 16..31 of the tile are zero)
    S_S0, S_L1, S_S1, S_L2, S_S2, S_L3, S_S3, S_LSTM0, S_LSTM1, S_HEADB, S_COUNT
};
// Channel order of the 16-STREAM kernel's STFT (silero_v4_t16.hip; the 32-stream kernel keeps v5::bin_of_channel): the order of
// the once-more-folded DFT, v5::bin_of_channel_fold3
__host__ __device__ constexpr int bin_of_channel_t16(int ch) { return v5::bin_of_channel_fold3(ch); }
constexpr int MAG_Q = 33;                      // quads per STFT column: 128 bins + Nyquist (+3 pad channels)
constexpr int MAG_ROWS = 8 * MAG_Q;            // 264 rows per tile, row = 33 t + q
// STFT part, LDS: reflect-padded frame [32][704] f32 + the 4-way fold of two columns (2 x 64 quad rows: pe, po, qe, qo as in
// V5) + window table w[256] + |X0|, |X128| of the two columns [2][2][32] + fold corrections [2][3][32].
// The two REAL bins (k = 0, k = 128) are summed in float64 from the samples: they are the graph's ill-conditioned
// inputs (log(1 + |X| 2^20) of a real sum that may cancel to ~1e-6 of its terms; a complex bin needs re AND im to
// cancel, which is ~100 x rarer) - DESIGN.md §3 "Numerics".
constexpr int K1_XP_QUADS = 176;               // reflect-padded frame: 704 samples per stream
constexpr int K1_XS_F4 = 32 * K1_XP_QUADS;
constexpr int K1_UV_ROWS = 128;
constexpr int K1_WT_F4 = 64;                   // w[n] as 64 quads, 256-byte aligned (the readers XOR-swizzle the quad index)
constexpr int K1_LDS_F4 = K1_XS_F4 + K1_UV_ROWS * QS + K1_WT_F4 + 32 + 48;
static_assert((K1_XS_F4 + K1_UV_ROWS * QS) % 16 == 0, "window table must start on a 16-quad boundary");
// tail, LDS rows
constexpr int R_A16 = MAG_ROWS;                // first-layer output: 264 + 4 t' + quad
constexpr int K2_ROWS = MAG_ROWS + 16;
constexpr int R_Y0 = 0, R_Y1 = 16, R_Y2 = 48, R_Y3 = 64, R_Y4 = 80, R_Y5 = 88, R_Y6 = 104;
constexpr int R_H0 = 120, R_H1 = 136, R_H0N = 152;   // h_{t-1} of LSTM layers 0 / 1, layer 0's new h
constexpr int K2_MISC_FLOATS = 32 + 8 * 32 + 256;  // mm[32], colmean[8][32], head partials [2 steps][4 waves][32]
// 8 kHz sub-model: two columns survive the third stride conv (stride 1); rows re-used from dead activations
constexpr int R8_Y4 = 80, R8_Y5 = 0, R8_Y6 = 32, R_H1N = 96;
constexpr int K2_LDS_F4 = K2_ROWS * QS + K2_MISC_FLOATS / 4 + 256;   // + partial log sums [4 waves][8 columns][32 streams]
constexpr int V4_SM_F4 = K1_LDS_F4 > K2_LDS_F4 ? K1_LDS_F4 : K2_LDS_F4;   // the tile's 32 state machines (96 B each) sit past both layouts
constexpr int V4_LDS_F4 = V4_SM_F4 + 192;
static_assert(V4_LDS_F4 * 16 <= 160 * 1024, "V4 LDS layout exceeds one CU");
}  // namespace v4

// per-slot hysteresis state (VADProcessor fields, core/silero_model.py:596-639), 96 bytes.
// Thresholds are doubles: the reference compares Python floats (float(np.float32 p) >= 0.7).
struct SmSlot {
    double start_prob, end_prob, start_ratio, end_ratio;
    int32_t start_count, end_count;
    int32_t active, n_start, n_end;
    int32_t start_len;       // len(recent_start_frames), deque maxlen 20
    uint32_t start_hist;     // its contents, newest in bit 0
    int32_t end_len;         // len(recent_end_frames), deque maxlen 100
    uint32_t end_hist[4];    // newest in bit 0 of word 0
    int32_t buffered;        // len(voice_buffer) in frames
    int32_t seg_frames;      // frames in current_voice_data (-1 = None)
    int32_t pad[2];
};
static_assert(sizeof(SmSlot) == 96, "SmSlot layout");

struct StepParams {
    const float *wstream;          // packed weight streams
    uint32_t wstream_bytes;
    uint32_t sect[NWAVES][16];     // block offset of each section, per wave
    float *state;                  // [max_streams][256]
    SmSlot *sm;                    // [max_streams]
    const int32_t *slots;          // [n] or nullptr (identity)
    const void *frames;            // [n][T][512]
    float *probs;                  // [n][T]
    uint8_t *events;               // [n][T] or nullptr
    int32_t *seg_frames;           // [n] or nullptr (last END of the call)
    int32_t n;
    int32_t T;                     // frames per stream in this call
    int32_t fmt;                   // vad_frame_format
    float thresh;                  // denoise gate, < 0 = off
    int32_t variant;               // 1 = the graph's 8 kHz sub-model (pack_weights.h): V4 two LSTM steps per frame, V5 256-sample frames
#ifdef VADK_STAMPS
    unsigned long long *stamps;    // diagnostic builds only (tools/kbench.cpp): [block][wave][16] s_memtime stamps
#endif
};

// resampler launch parameters (csrc/resample.hip)
struct ResampleSeg {
    const float *wstream;     // folded operator (pack_weights.cpp: pack_resample_operator)
    uint32_t wstream_bytes;
    uint32_t tile_blocks;     // blocks per 32-row tile = 8 + (n_in / 32) * 4
    uint32_t row128_block;    // first block of the VALU row's coefficients (outputs 128 / 384)
    const float *in;          // [n][n_in]
    float *out;               // [n][512]
    int32_t n;
    int32_t n_in;             // multiple of 256
};
// one launch resamples up to 4 segments (e.g. the 8 / 24 / 48 kHz clients of a tick): workgroups
// tile_start[s] .. tile_start[s+1]-1 serve segment s
constexpr int RESAMPLE_MAX_SEGS = 4;
struct ResampleParams {
    ResampleSeg seg[RESAMPLE_MAX_SEGS];
    int32_t nseg;
    int32_t tile_start[RESAMPLE_MAX_SEGS + 1];
};

// fused resample -> Silero V5 step on 16-stream tiles (csrc/silero_v5_t16.hip, RS instantiation): one launch for a tick whose
// streams arrive at different rates.  Segment k = n streams of one input rate; stream0 = index of its first stream in the
// call's slots / probs / events arrays.  The tiles walk the segments back to back in the order given here: tile b carries the
// "virtual" streams 16 b .. 16 b + 15 of that concatenation (vstart = where the segment begins in it), so a tile may hold the
// last streams of one segment and the first of the next - it then resamples each part with its own operator - and 4 096
// streams in three uneven thirds are exactly 256 tiles, one per CU.
struct RateSeg {
    const float *wstream;     // pack_resample_operator_t16; nullptr: the segment is 16 kHz already (`in` holds 512-sample frames)
    uint32_t wstream_bytes;
    uint32_t wave_blocks;     // operator blocks per wave
    uint32_t row128_block;
    int32_t n, n_in, stream0, vstart;
    const float *in;          // [n][n_in]
};
constexpr int RATE_MAX_SEGS = 8;
struct RateParams {
    RateSeg seg[RATE_MAX_SEGS];
    int32_t nseg;
    int32_t total;            // streams of all segments
};

}  // namespace vadk

// Host-side weight packing: SVW blob (canonical tensors) -> per-wave MFMA weight streams.
//
// The reference keeps Silero's weights inside the .onnx file and lets onnxruntime lay them out
// (/root/reference/src/real_time_vad/core/silero_model.py:321-325).  Here every wave of the
// fused kernel reads its weights as one linear stream of 1 KiB blocks (64 lanes x float4) in
// consumption order; see vad_layout.h for the fragment convention this code must mirror:
//   weight block (tile nt, k-iteration j):  lane (n' = l&31, hh = l>>5), component i
//       = W[32*nt + n'][channel 8j + 4hh + i]
//   bias / head block g:                    lane (any, hh), component i = b[32*nt + 8g + 4hh + i]
#include "pack_weights.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace vadk {

namespace {

struct SvwEntry {
    char name[48];
    uint32_t ndim, dims[4], reserved;
    uint64_t offset, nelem;
};
static_assert(sizeof(SvwEntry) == 88, "SVW entry layout (cutter_vad_amd/weights_io.py)");

struct Blob {
    const uint8_t *p;
    size_t len;
    uint32_t version, n;
    const SvwEntry *tab;
    std::string *err;

    const float *get(const char *name, uint64_t expect) const {
        for (uint32_t i = 0; i < n; ++i) {
            if (std::strncmp(tab[i].name, name, sizeof tab[i].name) == 0) {
                // overflow-safe range check: offset and nelem come from the file (an offset near 2^64 must not wrap past `len`)
                if (tab[i].nelem != expect || tab[i].offset > len || tab[i].nelem > (len - tab[i].offset) / 4) break;
                if (tab[i].offset % 4 != 0) break;      // tensors are read as float*
                return reinterpret_cast<const float *>(p + tab[i].offset);
            }
        }
        if (err->empty()) *err = std::string("Failed to load model: tensor '") + name + "' missing or of the wrong size";
        return nullptr;
    }
};

bool open_blob(const void *data, size_t len, Blob &b, std::string &err) {
    b.p = static_cast<const uint8_t *>(data);
    b.len = len;
    b.err = &err;
    if (!data || len < 16 || std::memcmp(data, "SVADW001", 8) != 0) {
        err = "Failed to load model: not an SVW weight blob (bad magic)";
        return false;
    }
    std::memcpy(&b.version, b.p + 8, 4);
    std::memcpy(&b.n, b.p + 12, 4);
    if (16 + (size_t)b.n * sizeof(SvwEntry) > len) {
        err = "Failed to load model: truncated SVW tensor table";
        return false;
    }
    b.tab = reinterpret_cast<const SvwEntry *>(b.p + 16);
    return true;
}

class StreamBuilder {
   public:
    std::vector<float> data;
    float *new_block() {
        data.resize(data.size() + BLK_FLOATS, 0.f);
        return data.data() + data.size() - BLK_FLOATS;
    }
    uint32_t blocks() const { return (uint32_t)(data.size() / BLK_FLOATS); }

    // conv / linear weight block: rows = output channels of tile `row0`.., k = input channel
    template <class F>
    void weight_block(F &&value /* (row n', channel c) -> float */, int j) {
        float *b = new_block();
        for (int l = 0; l < 64; ++l) {
            const int np = l & 31, hh = l >> 5;
            for (int i = 0; i < 4; ++i) b[l * 4 + i] = value(np, 8 * j + 4 * hh + i);
        }
    }
    // 16-row tile for v_mfma_f32_16x16x4_f32: lane l = (row l & 15, k group l >> 4), component i = input channel 16 j + 4 (l >> 4) + i
    template <class F>
    void weight_block16(F &&value /* (row r < 16, channel c) -> float */, int j) {
        float *b = new_block();
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) b[l * 4 + i] = value(l & 15, 16 * j + 4 * (l >> 4) + i);
    }
    // a per-output-channel vector in that tile's D layout: lane (column l & 15, l >> 4) holds channels 4 (l >> 4) .. + 3
    template <class F>
    void vector_block16(F &&value /* (channel < 16) -> float */) {
        float *b = new_block();
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) b[l * 4 + i] = value(4 * (l >> 4) + i);
    }
    // 4 lane-expanded blocks holding a per-output-channel vector for one 32-channel tile
    template <class F>
    void vector_blocks(F &&value /* (channel within tile) -> float */) {
        for (int g = 0; g < 4; ++g) {
            float *b = new_block();
            for (int l = 0; l < 64; ++l) {
                const int hh = l >> 5;
                for (int i = 0; i < 4; ++i) b[l * 4 + i] = value(8 * g + 4 * hh + i);
            }
        }
    }
};

// the folded STFT relies on exact symmetries of the stored basis; refuse weights that lack them
// N = window length (256; 128 for Silero V5's 8 kHz sub-model): rows 0..N/2 cosine, N/2+1..N+1 minus sine
bool check_stft_symmetry(const float *stft, std::string &err, int N = 256) {
    const int H = N / 2;
    for (int k = 0; k <= H; ++k) {
        const float *c = stft + (size_t)k * N, *sn = stft + (size_t)(H + 1 + k) * N;
        bool ok = c[0] == 0.f && sn[0] == 0.f && sn[H] == 0.f;
        for (int n = 1; n < H && ok; ++n) ok = c[n] == c[N - n] && sn[n] == -sn[N - n];
        if (k == 0 || k == H)
            for (int n = 0; n < N && ok; ++n) ok = sn[n] == 0.f;
        if (!ok) {
            err = "Failed to load model: STFT basis is not the symmetric windowed DFT the kernel assumes";
            return false;
        }
    }
    return true;
}

// The 4-way folded DFT (vad_layout.h, v5; V4 uses the same) uses analytic cos/sin and the window taken from the k = 0 row: the stored
// basis must BE that windowed DFT (it is for Silero: max |stored - w cos| = 5.7e-8)
bool check_windowed_dft(const float *stft, std::string &err, int N = 256) {
    const double two_pi = 6.283185307179586476925286766559;
    const int H = N / 2;
    double worst = 0;
    for (int k = 0; k <= H; ++k)
        for (int n = 0; n < N; ++n) {
            const double ph = two_pi * (double)((k * n) & (N - 1)) / (double)N, w = stft[n];
            worst = std::max(worst, std::fabs((double)stft[(size_t)k * N + n] - w * std::cos(ph)));
            worst = std::max(worst, std::fabs((double)stft[(size_t)(H + 1 + k) * N + n] + w * std::sin(ph)));
        }
    if (worst > 2e-7) {
        err = "Failed to load model: STFT basis is not a windowed DFT (the kernels evaluate it as a folded DFT)";
        return false;
    }
    return true;
}

// silero_v4_t16.hip folds the even bins k = 2 m once more, about n = 32: the operand of bin 2 m pairs pe[n] with (-1)^m pe[64 - n]
// and qe[n] with -(-1)^m qe[64 - n], n = 1..31.  That is exact if and only if the tables the packer generates have
// cos(2 pi (2 m) (64 - n) / 256) = (-1)^m cos(2 pi (2 m) n / 256) and sin(2 pi (2 m) (64 - n) / 256) = -(-1)^m sin(2 pi (2 m) n / 256),
// and the unpaired n = 32 has cos = 0 for m odd, sin = 0 for m even: checked on the very values the blocks are built from (the
// stored basis is tied to these analytic tables by check_windowed_dft)
bool check_even_bin_fold(std::string &err) {
    const double two_pi = 6.283185307179586476925286766559;
    double worst = 0;
    for (int m = 0; m < 64; ++m) {
        const int k = 2 * m;
        const double sg = (m & 1) ? -1.0 : 1.0;
        for (int n = 1; n < 32; ++n) {
            const double a = two_pi * (double)((k * n) & 255) / 256.0, b = two_pi * (double)((k * (64 - n)) & 255) / 256.0;
            worst = std::max(worst, std::fabs(std::cos(b) - sg * std::cos(a)));
            worst = std::max(worst, std::fabs(std::sin(b) + sg * std::sin(a)));
        }
        const double c32 = std::cos(two_pi * (double)((k * 32) & 255) / 256.0), s32 = std::sin(two_pi * (double)((k * 32) & 255) / 256.0);
        worst = std::max(worst, (m & 1) ? std::fabs(c32) : std::fabs(s32));
    }
    if (worst > 1e-12) {
        err = "Failed to load model: the DFT tables lack the even-bin symmetry about n = 32 that the 16-stream V4 kernel folds on";
        return false;
    }
    return true;
}

// cos / -sin blocks of the 4-way folded DFT: wave w owns the 32 bins bin_of_channel(32 w + r); k-iteration j
// contracts n = 8j .. 8j+7 (n = 0 is an unused slot: weight 0)
// the same for a window of 128 (Silero V5's 8 kHz sub-model), as 16-row tiles for v_mfma_f32_16x16x4_f32 so that all four waves
// share the 64 complex bins: wave w owns the bins 2 (16 (w & 1) + r) + (w >> 1), r = 0..15 (channels 16 w + r: parity w >> 1, the
// channel order of bin_of_channel_8k); two k-iterations of 16 folded samples, {cos, -sin} each
void pack_dft4_wave_128_t16(StreamBuilder &sb, int w) {
    const double two_pi = 6.283185307179586476925286766559;
    const int par = w >> 1, hf = w & 1;
    for (int j = 0; j < 2; ++j) {
        sb.weight_block16([&](int r, int n) { const int k = 2 * (16 * hf + r) + par; return n == 0 ? 0.f : (float)std::cos(two_pi * (double)((k * n) & 127) / 128.0); }, j);
        sb.weight_block16([&](int r, int n) { const int k = 2 * (16 * hf + r) + par; return n == 0 ? 0.f : (float)-std::sin(two_pi * (double)((k * n) & 127) / 128.0); }, j);
    }
}

// STFT blocks of the once-more-folded DFT (vad_layout.h, v5::bin_of_channel_fold3), 16-row tiles for v_mfma_f32_16x16x4_f32.
// Row tile 0 = 16 odd bins against the 4-way folded operands po | qo, K = 64: k-iterations j = 0..3 (n = 16 j .. 16 j + 15; n = 0 is
// an unused slot), {cos, -sin} each.  Row tile 1 = 16 even bins k = 2 m against the operands folded once more about n = 32
// (cos(2 pi m (64 - n) / 128) = (-1)^m cos(2 pi m n / 128), the sine with the opposite sign), K = 32: k-iterations j = 0, 1
// (n = 16 j .. 16 j + 15), {cos, -sin} each.  Slot n = 0 of those operands carries the sample the second fold cannot pair,
// n = 32: pe[32] for m even (weight cos(pi m / 2)), qe[32] for m odd (weight -sin(pi m / 2)); the other two are zero.
void pack_dft_fold3_wave(StreamBuilder &sb, int w) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < 4; ++j)
        for (int part = 0; part < 2; ++part)
            sb.weight_block16([&](int r, int n) {
                const int k = v5::bin_of_channel_fold3(32 * w + r);
                const double ph = two_pi * (double)((k * n) & 255) / 256.0;
                return n == 0 ? 0.f : (float)(part == 0 ? std::cos(ph) : -std::sin(ph));
            }, j);
    for (int j = 0; j < 2; ++j)
        for (int part = 0; part < 2; ++part)
            sb.weight_block16([&](int r, int n) {
                const int k = v5::bin_of_channel_fold3(32 * w + 16 + r), m = k / 2;
                const double ph = two_pi * (double)((k * (n == 0 ? 32 : n)) & 255) / 256.0;
                if (n == 0 && (part == 0) != (m % 2 == 0)) return 0.f;       // pe[32] enters the even m, qe[32] the odd m
                return (float)(part == 0 ? std::cos(ph) : -std::sin(ph));
            }, j);
}

void pack_dft4_wave(StreamBuilder &sb, int w) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < 8; ++j) {
        sb.weight_block([&](int np, int n) { const int k = v5::bin_of_channel(32 * w + np); return n == 0 ? 0.f : (float)std::cos(two_pi * (double)((k * n) & 255) / 256.0); }, j);
        sb.weight_block([&](int np, int n) { const int k = v5::bin_of_channel(32 * w + np); return n == 0 ? 0.f : (float)-std::sin(two_pi * (double)((k * n) & 255) / 256.0); }, j);
    }
}

}  // namespace

bool pack_silero_v5(const void *blob, size_t len, PackedWeights &out, std::string &err) {
    using namespace v5;
    Blob B;
    if (!open_blob(blob, len, B, err)) return false;
    if (B.version != 5) {
        err = "Failed to load model: weight blob is not Silero V5";
        return false;
    }
    // the graph's 8 kHz sub-model (else-branch): window 128, 65 bins, encoder.0 65 -> 128, 256-sample frames
    bool k8 = false;
    for (uint32_t i = 0; i < B.n; ++i)
        if (std::strncmp(B.tab[i].name, "meta.variant", sizeof B.tab[i].name) == 0) {
            const float *v = B.get("meta.variant", 1);
            k8 = v && v[0] == 8000.0f;
        }
    out.variant = k8 ? 1 : 0;
    const int N = k8 ? 128 : 256, NB = N / 2 + 1;         // window, bins
    const float *stft = B.get("stft.basis", (uint64_t)2 * NB * N);
    const float *ew[4], *eb[4];
    static const int co[4] = {128, 64, 64, 128};
    const int ci[4] = {NB, 128, 64, 64};
    for (int i = 0; i < 4; ++i) {
        char nm[32];
        std::snprintf(nm, sizeof nm, "enc%d.w", i);
        ew[i] = B.get(nm, (uint64_t)co[i] * ci[i] * 3);
        std::snprintf(nm, sizeof nm, "enc%d.b", i);
        eb[i] = B.get(nm, co[i]);
    }
    const float *w_ih = B.get("lstm.w_ih", 512 * 128), *w_hh = B.get("lstm.w_hh", 512 * 128);
    const float *b_ih = B.get("lstm.b_ih", 512), *b_hh = B.get("lstm.b_hh", 512);
    const float *head_w = B.get("head.w", 128), *head_b = B.get("head.b", 1);
    if (!err.empty()) return false;
    if (!check_stft_symmetry(stft, err, N)) return false;
    if (!check_windowed_dft(stft, err, N)) return false;
    if (!k8 && !check_even_bin_fold(err)) return false;

    StreamBuilder sb;
    auto convw = [&](int layer, int o, int c, int tap) -> float {
        return c < ci[layer] ? ew[layer][((size_t)o * ci[layer] + c) * 3 + tap] : 0.f;
    };
    for (int w = 0; w < NWAVES; ++w) {
        // STFT on the folded input (vad_layout.h): bins 32w..32w+31, {re, im} per k-iteration;
        // k-iteration j contracts n = 8j+1 .. 8j+8 (quad 2j on the lower half-wave, 2j+1 on the upper)
        out.sect[w][S_STFT] = sb.blocks();
        if (!k8) pack_dft_fold3_wave(sb, w);              // 128 bins: an odd and an even 16-row tile per wave, even bins folded once more
        else pack_dft4_wave_128_t16(sb, w);               // 64 bins = four 16-row tiles, one per wave
        // enc0 as a Toom-3 product (vad_layout.h): out channels 32w.., per k-iteration the five point-wise weight blocks
        // V(0) = w2, V(1)/2, V(-1)/2, V(2), V(inf) = w0 of V(z) = w2 + w1 z + w0 z^2 (evaluated in double); then the
        // Nyquist input channel: block A = points 0, 1, -1, 2 in the four components (lower half-wave), block B = inf
        out.sect[w][S_ENC0] = sb.blocks();
        sb.vector_blocks([&](int c) { return eb[0][32 * w + c]; });
        auto toom = [&](int o, int c129, int p) -> float {
            const double v0 = ew[0][((size_t)o * NB + c129) * 3 + 2], v1 = ew[0][((size_t)o * NB + c129) * 3 + 1],
                         v2 = ew[0][((size_t)o * NB + c129) * 3 + 0];
            switch (p) {
                case 0: return (float)v0;
                case 1: return (float)(0.5 * (v0 + v1 + v2));
                case 2: return (float)(0.5 * (v0 - v1 + v2));
                case 3: return (float)(v0 + 2.0 * v1 + 4.0 * v2);
                default: return (float)v2;
            }
        };
        for (int j = 0; j < (k8 ? 8 : 16); ++j)
            for (int p = 0; p < 5; ++p)
                sb.weight_block([&](int np, int c) { return toom(32 * w + np, k8 ? bin_of_channel_8k(c) : bin_of_channel_fold3(c), p); }, j);
        {
            float *a = sb.new_block();
            for (int l = 0; l < 32; ++l)
                for (int p = 0; p < 4; ++p) a[l * 4 + p] = toom(32 * w + l, NB - 1, p);
            float *b = sb.new_block();
            for (int l = 0; l < 32; ++l) b[l * 4] = toom(32 * w + l, NB - 1, 4);
        }
        // enc1: n-tile w&1, output column w>>1; taps (1,2) for column 0, (0,1) for column 1
        out.sect[w][S_ENC1] = sb.blocks();
        {
            const int nt = w & 1, tp = w >> 1;
            sb.vector_blocks([&](int c) { return eb[1][32 * nt + c]; });
            for (int ti = 0; ti < 2; ++ti) {
                const int tap = (1 - tp) + ti;
                for (int j = 0; j < 16; ++j)
                    sb.weight_block([&](int np, int c) { return convw(1, 32 * nt + np, c, tap); }, j);
            }
        }
        // enc2: n-tile w&1, taps 1,2 on input columns 0,1; waves 2,3 contract the second K half (column 1) of the
        // SAME streams (their section offsets alias waves 0,1 below)
        out.sect[w][S_ENC2] = sb.blocks();
        if (w < 2) {
            sb.vector_blocks([&](int c) { return eb[2][32 * w + c]; });
            for (int ti = 0; ti < 2; ++ti)
                for (int j = 0; j < 8; ++j)
                    sb.weight_block([&](int np, int c) { return convw(2, 32 * w + np, c, 1 + ti); }, j);
        }
        // enc3: n-tile w, centre tap only (single input column)
        out.sect[w][S_ENC3] = sb.blocks();
        sb.vector_blocks([&](int c) { return eb[3][32 * w + c]; });
        for (int j = 0; j < 8; ++j) sb.weight_block([&](int np, int c) { return convw(3, 32 * w + np, c, 1); }, j);
        // LSTM: hidden units 32w.., gates i,f,g,o (PyTorch row order), then the head weights
        out.sect[w][S_LSTM] = sb.blocks();
        for (int q = 0; q < 4; ++q)
            sb.vector_blocks([&](int c) { const int r = q * 128 + 32 * w + c; return b_ih[r] + b_hh[r]; });
        for (int j = 0; j < 16; ++j)
            for (int q = 0; q < 4; ++q)
                sb.weight_block([&](int np, int c) { return w_ih[(size_t)(q * 128 + 32 * w + np) * 128 + c]; }, j);
        for (int j = 0; j < 16; ++j)
            for (int q = 0; q < 4; ++q)
                sb.weight_block([&](int np, int c) { return w_hh[(size_t)(q * 128 + 32 * w + np) * 128 + c]; }, j);
        sb.vector_blocks([&](int c) { return head_w[32 * w + c]; });
        // block 148 of the section: the same gate biases once more, compact - floats [gate q][unit c] - for the kernel that keeps
        // them in LDS and initialises its accumulators from there (silero_v5.hip: 512 B per wave instead of 16 KB at the head
        // of the launch's first requests)
        {
            float *cb = sb.new_block();
            for (int q = 0; q < 4; ++q)
                for (int c = 0; c < 32; ++c) cb[q * 32 + c] = b_ih[q * 128 + 32 * w + c] + b_hh[q * 128 + 32 * w + c];
        }
    }
    out.sect[2][S_ENC2] = out.sect[0][S_ENC2];
    out.sect[3][S_ENC2] = out.sect[1][S_ENC2];
    const uint32_t hb = sb.blocks();
    sb.new_block()[0] = head_b[0];
    // bin 128 of the folded STFT (VALU): floats 0..127 = C[128][1..128]; its sine row is exactly zero
    // the window of the stored basis: its k = 0 cosine row
    const uint32_t nb = sb.blocks();
    std::memcpy(sb.new_block(), stft, (size_t)N * sizeof(float));
    for (int w = 0; w < NWAVES; ++w) {
        out.sect[w][S_HEADB] = hb;
        out.sect[w][S_NYQ] = nb;
    }
    out.data = std::move(sb.data);
    return true;
}

// ---- Silero V5 16 kHz for the 16-stream tile kernel (csrc/silero_v5_t16.hip) -------------------------------------------------
// Same sections and the same algebra (4-way folded DFT, Toom-3 enc0) as pack_silero_v5, but every contraction runs on
// v_mfma_f32_16x16x4_f32: a wave's 32 output channels are TWO 16-row tiles rt = 0, 1, a k-iteration j contracts 16 channels, and a
// 1 KiB block holds 16 rows x 16 channels: lane l = (row l & 15, channel group kq = l >> 4), component i = channel 16 j + 4 kq + i
// (StreamBuilder::weight_block16).  A per-channel vector of a tile is ONE block in the tile's D layout (vector_block16).
bool pack_silero_v5_t16(const void *blob, size_t len, PackedWeights &out, std::string &err) {
    using namespace v5;
    Blob B;
    if (!open_blob(blob, len, B, err)) return false;
    if (B.version != 5) {
        err = "Failed to load model: weight blob is not Silero V5";
        return false;
    }
    // the graph's 8 kHz sub-model (else-branch): window 128, 65 bins, encoder.0 65 -> 128, 256-sample frames
    bool k8 = false;
    for (uint32_t i = 0; i < B.n; ++i)
        if (std::strncmp(B.tab[i].name, "meta.variant", sizeof B.tab[i].name) == 0) {
            const float *v = B.get("meta.variant", 1);
            k8 = v && v[0] == 8000.0f;
        }
    const int N = k8 ? 128 : 256, NB = N / 2 + 1;
    const float *stft = B.get("stft.basis", (uint64_t)2 * NB * N);
    const float *ew[4], *eb[4];
    static const int co[4] = {128, 64, 64, 128};
    const int ci[4] = {NB, 128, 64, 64};
    for (int i = 0; i < 4; ++i) {
        char nm[32];
        std::snprintf(nm, sizeof nm, "enc%d.w", i);
        ew[i] = B.get(nm, (uint64_t)co[i] * ci[i] * 3);
        std::snprintf(nm, sizeof nm, "enc%d.b", i);
        eb[i] = B.get(nm, co[i]);
    }
    const float *w_ih = B.get("lstm.w_ih", 512 * 128), *w_hh = B.get("lstm.w_hh", 512 * 128);
    const float *b_ih = B.get("lstm.b_ih", 512), *b_hh = B.get("lstm.b_hh", 512);
    const float *head_w = B.get("head.w", 128), *head_b = B.get("head.b", 1);
    if (!err.empty()) return false;
    if (!check_stft_symmetry(stft, err, N)) return false;
    if (!check_windowed_dft(stft, err, N)) return false;
    if (!k8 && !check_even_bin_fold(err)) return false;
    out.variant = k8 ? 1 : 0;
    StreamBuilder sb;
    auto convw = [&](int layer, int o, int c, int tap) -> float { return ew[layer][((size_t)o * ci[layer] + c) * 3 + tap]; };
    auto toom = [&](int o, int cin, int p) -> float {
        const double v0 = ew[0][((size_t)o * NB + cin) * 3 + 2], v1 = ew[0][((size_t)o * NB + cin) * 3 + 1],
                     v2 = ew[0][((size_t)o * NB + cin) * 3 + 0];
        switch (p) {
            case 0: return (float)v0;
            case 1: return (float)(0.5 * (v0 + v1 + v2));
            case 2: return (float)(0.5 * (v0 - v1 + v2));
            case 3: return (float)(v0 + 2.0 * v1 + 4.0 * v2);
            default: return (float)v2;
        }
    };
    for (int w = 0; w < NWAVES; ++w) {
        // STFT: the once-more-folded DFT (pack_dft_fold3_wave): row tile 0 = odd bins (4 k-iterations x {cos, -sin}), row tile 1 =
        // even bins on the operands folded again about n = 32 (2 k-iterations x {cos, -sin})
        out.sect[w][S_STFT] = sb.blocks();
        if (!k8) pack_dft_fold3_wave(sb, w);
        else pack_dft4_wave_128_t16(sb, w);               // 8 kHz: 64 bins = four 16-row tiles, one per wave (as the 32-stream kernel)
        // enc0 (Toom-3): bias rt0, rt1; per k-iteration the five points x two row tiles; Nyquist channel: points 0,1,-1,2 in the
        // components of lanes kq = 0 (the other channel groups carry zeros), rt0, rt1; then the point at infinity, rt0, rt1
        out.sect[w][S_ENC0] = sb.blocks();
        for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return eb[0][32 * w + 16 * rt + c]; });
        for (int j = 0; j < (k8 ? 4 : 8); ++j)
            for (int p = 0; p < 5; ++p)
                for (int rt = 0; rt < 2; ++rt)
                    sb.weight_block16([&](int r, int c) { return toom(32 * w + 16 * rt + r, k8 ? bin_of_channel_8k(c) : bin_of_channel_fold3(c), p); }, j);
        for (int part = 0; part < 2; ++part)
            for (int rt = 0; rt < 2; ++rt) {
                float *a = sb.new_block();
                for (int r = 0; r < 16; ++r)
                    for (int p = 0; p < (part == 0 ? 4 : 1); ++p) a[r * 4 + p] = toom(32 * w + 16 * rt + r, NB - 1, part == 0 ? p : 4);
            }
        // enc1: n-tile w&1, output column w>>1; taps (1,2) for column 0, (0,1) for column 1; iteration it = 8 ti + j
        out.sect[w][S_ENC1] = sb.blocks();
        {
            const int nt = w & 1, tp = w >> 1;
            for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return eb[1][32 * nt + 16 * rt + c]; });
            for (int ti = 0; ti < 2; ++ti)
                for (int j = 0; j < 8; ++j)
                    for (int rt = 0; rt < 2; ++rt)
                        sb.weight_block16([&](int r, int c) { return convw(1, 32 * nt + 16 * rt + r, c, (1 - tp) + ti); }, j);
        }
        // enc2: n-tile w&1 (packed for waves 0,1; waves 2,3 read the second K half of the same streams), taps 1,2 on columns 0,1
        out.sect[w][S_ENC2] = sb.blocks();
        if (w < 2) {
            for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return eb[2][32 * w + 16 * rt + c]; });
            for (int ti = 0; ti < 2; ++ti)
                for (int j = 0; j < 4; ++j)
                    for (int rt = 0; rt < 2; ++rt)
                        sb.weight_block16([&](int r, int c) { return convw(2, 32 * w + 16 * rt + r, c, 1 + ti); }, j);
        }
        // enc3: centre tap
        out.sect[w][S_ENC3] = sb.blocks();
        for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return eb[3][32 * w + 16 * rt + c]; });
        for (int j = 0; j < 4; ++j)
            for (int rt = 0; rt < 2; ++rt)
                sb.weight_block16([&](int r, int c) { return convw(3, 32 * w + 16 * rt + r, c, 1); }, j);
        // LSTM: biases (gate q, rt), W_ih and W_hh per k-iteration (gate q, rt), head weights (rt)
        out.sect[w][S_LSTM] = sb.blocks();
        for (int q = 0; q < 4; ++q)
            for (int rt = 0; rt < 2; ++rt)
                sb.vector_block16([&](int c) { const int r = q * 128 + 32 * w + 16 * rt + c; return b_ih[r] + b_hh[r]; });
        for (const float *W : {w_ih, w_hh})
            for (int j = 0; j < 8; ++j)
                for (int q = 0; q < 4; ++q)
                    for (int rt = 0; rt < 2; ++rt)
                        sb.weight_block16([&](int r, int c) { return W[(size_t)(q * 128 + 32 * w + 16 * rt + r) * 128 + c]; }, j);
        for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return head_w[32 * w + 16 * rt + c]; });
        // block 138 of the section: the gate biases once more, compact - floats [gate q][unit] (silero_v5_t16.hip keeps them in LDS)
        {
            float *cb = sb.new_block();
            for (int q = 0; q < 4; ++q)
                for (int c = 0; c < 32; ++c) cb[q * 32 + c] = b_ih[q * 128 + 32 * w + c] + b_hh[q * 128 + 32 * w + c];
        }
    }
    out.sect[2][S_ENC2] = out.sect[0][S_ENC2];
    out.sect[3][S_ENC2] = out.sect[1][S_ENC2];
    const uint32_t hb = sb.blocks();
    sb.new_block()[0] = head_b[0];
    const uint32_t nb = sb.blocks();
    std::memcpy(sb.new_block(), stft, (size_t)N * sizeof(float));
    for (int w = 0; w < NWAVES; ++w) {
        out.sect[w][S_HEADB] = hb;
        out.sect[w][S_NYQ] = nb;
    }
    out.data = std::move(sb.data);
    return true;
}

namespace {
void build_resample_operator_d(int n_in, std::vector<double> &R);
}

void build_resample_operator(int n_in, std::vector<float> &R) {
    std::vector<double> Rd;
    build_resample_operator_d(n_in, Rd);
    R.resize(Rd.size());
    for (size_t i = 0; i < Rd.size(); ++i) R[i] = (float)Rd[i];
}

namespace {
void build_resample_operator_d(int n_in, std::vector<double> &R) {
    // scipy.signal.resample, real input, window=None:
    //   X = rfft(x); Y[0:N/2+1] = X[0:N/2+1] with N = min(n_in, n_out); if N is even the bin N/2 is
    //   doubled when down-sampling and halved when up-sampling; y = irfft(Y, n_out) * n_out / n_in.
    // Hence y[o] = (1/n_in) * sum_i x[i] * ( D_M(theta) + nyquist term ),  theta = 2 pi (o/n_out - i/n_in),
    // D_M(theta) = 1 + 2 sum_{k=1..M} cos(k theta) = sin((M + 1/2) theta) / sin(theta / 2).
    const int n_out = 512;
    const int N = n_in < n_out ? n_in : n_out;
    const int M = (N % 2 == 0) ? N / 2 - 1 : (N - 1) / 2;
    const double two_pi = 6.283185307179586476925286766559;
    const long long period = (long long)n_in * n_out;
    R.assign((size_t)n_out * n_in, 0.0);
    for (int o = 0; o < n_out; ++o) {
        for (int i = 0; i < n_in; ++i) {
            long long p = ((long long)o * n_in - (long long)i * n_out) % period;   // exact phase reduction
            if (p < 0) p += period;
            const double theta = two_pi * (double)p / (double)period;
            const double sh = std::sin(0.5 * theta);
            double v = std::fabs(sh) < 1e-13 ? (double)(2 * M + 1) : std::sin((M + 0.5) * theta) / sh;
            if (N % 2 == 0 && n_in != n_out) {
                const int kq = N / 2;
                const double ci = std::cos(two_pi * (double)(((long long)kq * i) % n_in) / (double)n_in);
                const double co = std::cos(two_pi * (double)(((long long)kq * o) % n_out) / (double)n_out);
                // down: Y[kq] = 2 X[kq] and kq is the output Nyquist bin (irfft weight 1, real part only)
                // up  : Y[kq] = X[kq] / 2 (X[kq] is the real input Nyquist bin), a regular output bin (weight 2)
                v += (n_out < n_in) ? 2.0 * ci * co : ci * co;
            }
            R[(size_t)o * n_in + i] = v / (double)n_in;
        }
    }
}
}  // namespace

// The operator has two exact symmetries (both checked below; indices mod 512 / mod n):
//   half-period shift  R[o + 256][i + n/2] == R[o][i]      (a radix-2 step: even / odd harmonics separate)
//   mirror             R[512 - o][n - i]   == R[o][i]
// With H = n/2, Q = n/4, xe[i] = x[i] + x[i+H], xo[i] = x[i] - x[i+H], RE[o][i] = R[o][i] + R[o][i+H], RO = R - R(shifted):
//   y[o] + y[o+256] = sum_{i<H} RE[o][i] xe[i],   y[o] - y[o+256] = sum_{i<H} RO[o][i] xo[i]          (o < 256)
// and each of the two half-size products folds about its own midpoint (RO and xo are anti-periodic in H, which moves the
// unpaired j = 0 term from the symmetric to the antisymmetric part):
//   ue[j] = xe[j] + xe[H-j], ve[j] = xe[j] - xe[H-j], uo[j] = xo[j] + xo[H-j], vo[j] = xo[j] - xo[H-j]   (0 < j < Q)
//   ue[0] = xe[0], ve[0] = 0, uo[0] = 0, vo[0] = xo[0]
//   se[o] = GSE[o].ue + RE[o][Q] xe[Q] / 2,  ae[o] = GAE[o].ve,  so[o] = GSO[o].uo + RO[o][Q] xo[Q] / 2,  ao[o] = GAO[o].vo
//   G{S,A}{E,O}[o][j] = (R{E,O}[o][j] +/- R{E,O}[o][H-j]) / 4  (j > 0),  GSE[o][0] = RE[o][0] / 2,  GAO[o][0] = RO[o][0] / 2
//   y[o] = se+ae+so+ao,  y[o+256] = se+ae-so-ao,  y[256-o] = se-ae+so-ao,  y[512-o] = se-ae-so+ao        (o = 0..127)
//   y[128] = se[128] + so[128],  y[384] = se[128] - so[128]
// 4 x 128 rows of K = n/4 instead of 512 rows of K = n: a quarter of the dense MFMAs.
// Packed layout: row tile t (4 of them) = rows o = 32 t .. 32 t + 31:
//   4 vector blocks RE[o][Q] / 2, 4 vector blocks RO[o][Q] / 2, then per k-iteration (8 values of j) the blocks SE, AE, SO, AO;
// after the four tiles: GSE[128][0..Q), GSO[128][0..Q), RE[128][Q] / 2, RO[128][Q] / 2 as plain floats.

namespace {
// R[o][3 i'] == (512 / n) [o == m i'] + (-1)^(o - m i') / n  (m = 1536 / n) for n = 768, 1536?  (VAD_RS_DENSE: diagnostic - A/B and
// tests of the two operator layouts)
// n = 256 (8 kHz -> 16 kHz, up by exactly two): R[2 i][i'] == [i == i'] - every input sample IS an even output sample
bool resample_up2(int n, const std::vector<double> &R) {
    if (n != 256 || std::getenv("VAD_RS_DENSE")) return false;
    double dev = 0;
    for (int i = 0; i < 256; ++i)
        for (int ip = 0; ip < 256; ++ip) dev = std::max(dev, std::fabs(R[(size_t)(2 * i) * n + ip] - (i == ip ? 1.0 : 0.0)));
    return dev <= 1e-12;
}
bool resample_poly3(int n, const std::vector<double> &R) {
    if ((n != 768 && n != 1536) || std::getenv("VAD_RS_DENSE")) return false;
    const int m = 1536 / n;
    double dev = 0;
    for (int o = 0; o < 512; ++o)
        for (int ip = 0; ip < n / 3; ++ip) {
            const double want = ((o == m * ip) ? 512.0 / n : 0.0) + (((o - m * ip) & 1) ? -1.0 : 1.0) / n;
            dev = std::max(dev, std::fabs(R[(size_t)o * n + 3 * ip] - want));
        }
    return dev <= 1e-12;
}
}  // namespace

uint32_t pack_resample_operator(int n_in, std::vector<float> &out, uint32_t *row128_block, std::string &err) {
    std::vector<double> R;
    build_resample_operator_d(n_in, R);
    const int n = n_in, H = n / 2, Q = n / 4;
    if (n % 256) {
        err = "resample chunk length must be a multiple of 256 samples";
        return 0;
    }
    auto Rv = [&](int o, int i) { return R[(size_t)(o & 511) * n + (i % n)]; };
    double worst = 0;
    for (int o = 0; o < 512; ++o)
        for (int i = 0; i < n; ++i) {
            worst = std::max(worst, std::fabs(Rv(o, i) - Rv(512 - o, n - i)));
            worst = std::max(worst, std::fabs(Rv(o, i) - Rv(o + 256, i + H)));
        }
    if (worst > 1e-12) {
        err = "resample operator lacks the symmetries the folded kernel relies on";
        return 0;
    }
    // (the fused kernel's layout, pack_resample_operator_t16 below, leaves every third column of the 24 / 48 kHz operators out and
    // copies those samples.  Not here: this kernel's outputs go to HBM, the copied samples arrive chunk by chunk on other threads
    // than the ones that own the output rows, and the 66 KB of LDS that would carry them across would end the two-workgroups-
    // per-CU residency its small launches rely on.)
    const int Kc = Q;
    auto jm = [&](int i) { return i; };
    auto RE = [&](int o, int i) { return Rv(o, i) + Rv(o, i + H); };
    auto RO = [&](int o, int i) { return Rv(o, i) - Rv(o, i + H); };
    auto GSE = [&](int o, int j) -> float { return (float)(j == 0 ? 0.5 * RE(o, 0) : 0.25 * (RE(o, j) + RE(o, H - j))); };
    auto GAE = [&](int o, int j) -> float { return (float)(j == 0 ? 0.0 : 0.25 * (RE(o, j) - RE(o, H - j))); };
    auto GSO = [&](int o, int j) -> float { return (float)(j == 0 ? 0.0 : 0.25 * (RO(o, j) + RO(o, H - j))); };
    auto GAO = [&](int o, int j) -> float { return (float)(j == 0 ? 0.5 * RO(o, 0) : 0.25 * (RO(o, j) - RO(o, H - j))); };
    StreamBuilder sb;
    for (int t = 0; t < 4; ++t) {
        sb.vector_blocks([&](int c) { return (float)(0.5 * RE(32 * t + c, Q)); });
        sb.vector_blocks([&](int c) { return (float)(0.5 * RO(32 * t + c, Q)); });
        for (int j = 0; j < Kc / 8; ++j) {
            sb.weight_block([&](int np, int k) { return GSE(32 * t + np, jm(k)); }, j);
            sb.weight_block([&](int np, int k) { return GAE(32 * t + np, jm(k)); }, j);
            sb.weight_block([&](int np, int k) { return GSO(32 * t + np, jm(k)); }, j);
            sb.weight_block([&](int np, int k) { return GAO(32 * t + np, jm(k)); }, j);
        }
    }
    const uint32_t per_tile = sb.blocks() / 4;
    *row128_block = sb.blocks();
    std::vector<float> row(2 * (size_t)Kc + 2);
    for (int j = 0; j < Kc; ++j) {
        row[j] = GSE(128, jm(j));
        row[Kc + j] = GSO(128, jm(j));
    }
    row[2 * Kc] = (float)(0.5 * RE(128, Q));
    row[2 * Kc + 1] = (float)(0.5 * RO(128, Q));
    for (size_t j0 = 0; j0 < row.size(); j0 += BLK_FLOATS) {
        float *b = sb.new_block();
        for (size_t j = j0; j < j0 + BLK_FLOATS && j < row.size(); ++j) b[j - j0] = row[j];
    }
    out = std::move(sb.data);
    return per_tile;
}

// The same folded operator for the fused resample -> Silero V5 kernel on 16-stream tiles (silero_v5_t16.hip, RS = true):
// 16 x 16 x 4 MFMA tiles, wave w owns output rows o = 32 w .. 32 w + 31 (two row tiles rt) of all four parts.  Per wave:
//   4 vector blocks (D layout): RE[o][Q] / 2 for rt 0, 1, then RO[o][Q] / 2 for rt 0, 1;
//   per k-iteration (16 values of j): the blocks SE rt0, SE rt1, AE rt0, AE rt1, SO rt0, SO rt1, AO rt0, AO rt1;
// after the four waves the plain floats of row 128 as in pack_resample_operator.  Returns the blocks per wave.
uint32_t pack_resample_operator_t16(int n_in, std::vector<float> &out, uint32_t *row128_block, std::string &err) {
    std::vector<double> R;
    build_resample_operator_d(n_in, R);
    const int n = n_in, H = n / 2, Q = n / 4;
    if (n % 256) {
        err = "resample chunk length must be a multiple of 256 samples";
        return 0;
    }
    auto Rv = [&](int o, int i) { return R[(size_t)(o & 511) * n + (i % n)]; };
    double worst = 0;
    for (int o = 0; o < 512; ++o)
        for (int i = 0; i < n; ++i) {
            worst = std::max(worst, std::fabs(Rv(o, i) - Rv(512 - o, n - i)));
            worst = std::max(worst, std::fabs(Rv(o, i) - Rv(o + 256, i + H)));
        }
    if (worst > 1e-12) {
        err = "resample operator lacks the symmetries the folded kernel relies on";
        return 0;
    }
    // 24 / 48 kHz (n = 768 / 1536 = 3 n'): every third input sample sits ON an instant of the 512-sample output grid, m = 1536 / n
    // output steps apart, and there the Dirichlet kernel is a delta plus the Nyquist bin's alternation:
    //   R[o][3 i'] = (512 / n) [o == m i'] + (-1)^(o - m i') / n
    // - a third of the operator's columns are a copy, a scale and ONE alternating sum per chunk.  The kernel does exactly that
    // (silero_v5_t16.hip, "P3") and contracts only the folded samples j = 1, 2, 4, 5, 7, 8, ... : K = n / 6 per part instead of
    // n / 4.  Checked here against the operator itself; any other length keeps the full contraction.
    const bool poly3 = resample_poly3(n, R);
    const int Kc = poly3 ? 2 * Q / 3 : Q;                                   // contraction length per folded part
    auto jm = [&](int i) { return poly3 ? 3 * (i >> 1) + 1 + (i & 1) : i; };   // contraction index -> folded sample j
    auto RE = [&](int o, int i) { return Rv(o, i) + Rv(o, i + H); };
    auto RO = [&](int o, int i) { return Rv(o, i) - Rv(o, i + H); };
    auto G = [&](int part, int o, int j) -> float {
        switch (part) {
            case 0: return (float)(j == 0 ? 0.5 * RE(o, 0) : 0.25 * (RE(o, j) + RE(o, H - j)));   // GSE
            case 1: return (float)(j == 0 ? 0.0 : 0.25 * (RE(o, j) - RE(o, H - j)));             // GAE
            case 2: return (float)(j == 0 ? 0.0 : 0.25 * (RO(o, j) + RO(o, H - j)));             // GSO
            default: return (float)(j == 0 ? 0.5 * RO(o, 0) : 0.25 * (RO(o, j) - RO(o, H - j)));  // GAO
        }
    };
    // 8 kHz (n = 256, up by two): the even outputs are the input samples themselves (checked above all else by resample_up2),
    // and an output's parity is the parity of its folded row o - so only the ODD rows are contracted: per wave ONE 16-row tile,
    // rows 32 w + 2 r + 1, two vector blocks (RE, RO) and four weight blocks per k-iteration.  The kernel copies the samples
    // (silero_v5_t16.hip, part shape 3) and reads the layout off the block count.
    const bool up2 = resample_up2(n, R);
    StreamBuilder sb;
    for (int w = 0; w < 4; ++w) {
        if (up2) {
            sb.vector_block16([&](int c) { return (float)(0.5 * RE(32 * w + 2 * c + 1, Q)); });
            sb.vector_block16([&](int c) { return (float)(0.5 * RO(32 * w + 2 * c + 1, Q)); });
            for (int j = 0; j < Q / 16; ++j)
                for (int part = 0; part < 4; ++part)
                    sb.weight_block16([&](int r, int k) { return G(part, 32 * w + 2 * r + 1, k); }, j);
            continue;
        }
        for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return (float)(0.5 * RE(32 * w + 16 * rt + c, Q)); });
        for (int rt = 0; rt < 2; ++rt) sb.vector_block16([&](int c) { return (float)(0.5 * RO(32 * w + 16 * rt + c, Q)); });
        for (int j = 0; j < Kc / 16; ++j)
            for (int part = 0; part < 4; ++part)
                for (int rt = 0; rt < 2; ++rt)
                    sb.weight_block16([&](int r, int k) { return G(part, 32 * w + 16 * rt + r, jm(k)); }, j);
    }
    const uint32_t per_wave = sb.blocks() / 4;
    *row128_block = sb.blocks();
    std::vector<float> row(2 * (size_t)Kc + 2);
    for (int j = 0; j < Kc; ++j) {
        row[j] = G(0, 128, jm(j));
        row[Kc + j] = G(2, 128, jm(j));
    }
    row[2 * Kc] = (float)(0.5 * RE(128, Q));          // (with every third sample skipped the kernel does not use these two, nor
    row[2 * Kc + 1] = (float)(0.5 * RO(128, Q));      //  the vector blocks above: x[Q] and x[3Q] are such samples)
    for (size_t j0 = 0; j0 < row.size(); j0 += BLK_FLOATS) {
        float *b = sb.new_block();
        for (size_t j = j0; j < j0 + BLK_FLOATS && j < row.size(); ++j) b[j - j0] = row[j];
    }
    out = std::move(sb.data);
    return per_wave;
}

namespace {
// the tensors of a Silero V4 blob (either sub-model), by role
struct V4Tensors {
    const float *stft, *filt;
    const float *dww[4], *dwb[4], *pww[4], *pwb[4], *pjw[4], *pjb[4], *sw[4], *sbias[4];
    const float *lwi[2], *lwh[2], *lbi[2], *lbh[2], *head_w, *head_b;
    int variant;
};
constexpr int V4_CI[4] = {258, 16, 32, 32}, V4_CO[4] = {16, 32, 32, 64}, V4_SC[4] = {16, 32, 32, 64};

bool load_v4(const void *blob, size_t len, Blob &B, V4Tensors &t, std::string &err) {
    if (!open_blob(blob, len, B, err)) return false;
    if (B.version != 4) {
        err = "Failed to load model: weight blob is not Silero V4";
        return false;
    }
    t = V4Tensors{};
    t.stft = B.get("stft.basis", 258 * 256);
    t.filt = B.get("norm.filter", 7);
    char nm[32];
    for (int i = 0; i < 4; ++i) {
        std::snprintf(nm, sizeof nm, "l%d.dw.w", i); t.dww[i] = B.get(nm, (uint64_t)V4_CI[i] * 5);
        std::snprintf(nm, sizeof nm, "l%d.dw.b", i); t.dwb[i] = B.get(nm, V4_CI[i]);
        std::snprintf(nm, sizeof nm, "l%d.pw.w", i); t.pww[i] = B.get(nm, (uint64_t)V4_CO[i] * V4_CI[i]);
        std::snprintf(nm, sizeof nm, "l%d.pw.b", i); t.pwb[i] = B.get(nm, V4_CO[i]);
        if (i != 2) {
            std::snprintf(nm, sizeof nm, "l%d.proj.w", i); t.pjw[i] = B.get(nm, (uint64_t)V4_CO[i] * V4_CI[i]);
            std::snprintf(nm, sizeof nm, "l%d.proj.b", i); t.pjb[i] = B.get(nm, V4_CO[i]);
        }
        std::snprintf(nm, sizeof nm, "s%d.w", i); t.sw[i] = B.get(nm, (uint64_t)V4_SC[i] * V4_SC[i]);
        std::snprintf(nm, sizeof nm, "s%d.b", i); t.sbias[i] = B.get(nm, V4_SC[i]);
    }
    for (int l = 0; l < 2; ++l) {
        std::snprintf(nm, sizeof nm, "lstm%d.w_ih", l); t.lwi[l] = B.get(nm, 256 * 64);
        std::snprintf(nm, sizeof nm, "lstm%d.w_hh", l); t.lwh[l] = B.get(nm, 256 * 64);
        std::snprintf(nm, sizeof nm, "lstm%d.b_ih", l); t.lbi[l] = B.get(nm, 256);
        std::snprintf(nm, sizeof nm, "lstm%d.b_hh", l); t.lbh[l] = B.get(nm, 256);
    }
    t.head_w = B.get("head.w", 64);
    t.head_b = B.get("head.b", 1);
    if (!err.empty()) return false;
    {   // optional: meta.variant = 8000 marks the graph's else-branch (weights_io._extract_v4)
        std::string none;
        std::string *keep = B.err;
        B.err = &none;
        const float *var = B.get("meta.variant", 1);
        B.err = keep;
        t.variant = (var && var[0] == 8000.0f) ? 1 : 0;
    }
    if (!check_stft_symmetry(t.stft, err)) return false;
    if (!check_windowed_dft(t.stft, err)) return false;
    return true;
}

// the STFT kernels emit their 128 regular bins in the even/odd order of the 4-way folded DFT (v5::bin_of_channel);
// the first layer's per-channel tables and weight columns follow that order.  Channel 128 (Nyquist) stays.
inline int v4_bin32(int c) { return c < 128 ? v5::bin_of_channel(c) : c; }

// depthwise taps (k = 0..4) + bias (k = 5) per channel quad as float4 rows: row = (q * 6 + k), value i = channel 4q + i
uint32_t v4_dw_table(StreamBuilder &sb, const float *w5, const float *b, int C, int nquads) {
    const uint32_t first = sb.blocks();
    std::vector<float> tab((size_t)nquads * 6 * 4, 0.f);
    for (int q = 0; q < nquads; ++q)
        for (int k = 0; k < 6; ++k)
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * q + i;
                if (c < C) tab[((size_t)q * 6 + k) * 4 + i] = k < 5 ? w5[(size_t)c * 5 + k] : b[c];
            }
    for (size_t off = 0; off < tab.size(); off += BLK_FLOATS) {
        float *blk = sb.new_block();
        std::memcpy(blk, tab.data() + off, sizeof(float) * std::min<size_t>(BLK_FLOATS, tab.size() - off));
    }
    return first;
}

// S_DW0 and S_L0, shared by both tile shapes (the first layer runs on 16 x 16 x 4 tiles in either kernel).
// S_DW0: part 0 = magnitude channels 0..128 (34 quads), part 1 = normalised channels 129..257, then the 7-tap filter.
// S_L0: 16 outputs.  Bias block, then the Nyquist channel's four weight columns (pw|mag, proj|mag, pw|norm, proj|norm; that
// channel is contracted on the VALU), then per k-iteration of 16 channels the same four operands.
void v4_first_layer(StreamBuilder &sb, const V4Tensors &t, uint32_t *sec, int (*v4_bin)(int)) {
    using namespace v4;
    sec[S_DW0] = sb.blocks();
    std::vector<float> tab((size_t)(2 * 34 * 6 + 2) * 4, 0.f);
    for (int p = 0; p < 2; ++p)
        for (int q = 0; q < 34; ++q)
            for (int k = 0; k < 6; ++k)
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * q + i;
                    if (c < 129)
                        tab[(((size_t)p * 34 + q) * 6 + k) * 4 + i] =
                            k < 5 ? t.dww[0][(size_t)(129 * p + v4_bin(c)) * 5 + k] : t.dwb[0][129 * p + v4_bin(c)];
                }
    for (int k = 0; k < 7; ++k) tab[(size_t)(2 * 34 * 6) * 4 + k] = t.filt[k];
    for (size_t off = 0; off < tab.size(); off += BLK_FLOATS) {
        float *blk = sb.new_block();
        std::memcpy(blk, tab.data() + off, sizeof(float) * std::min<size_t>(BLK_FLOATS, tab.size() - off));
    }
    sec[S_L0] = sb.blocks();
    sb.vector_block16([&](int r) { return t.pwb[0][r] + t.pjb[0][r]; });
    for (int p = 0; p < 2; ++p) {
        sb.vector_block16([&](int r) { return t.pww[0][(size_t)r * 258 + 129 * p + v4_bin(128)]; });
        sb.vector_block16([&](int r) { return t.pjw[0][(size_t)r * 258 + 129 * p + v4_bin(128)]; });
    }
    for (int j = 0; j < 8; ++j)
        for (int p = 0; p < 2; ++p) {
            sb.weight_block16([&](int r, int c) { return t.pww[0][(size_t)r * 258 + 129 * p + v4_bin(c)]; }, j);
            sb.weight_block16([&](int r, int c) { return t.pjw[0][(size_t)r * 258 + 129 * p + v4_bin(c)]; }, j);
        }
}
}  // namespace

bool pack_silero_v4(const void *blob, size_t len, PackedWeights &out, std::string &err) {
    using namespace v4;
    Blob B;
    V4Tensors t;
    if (!load_v4(blob, len, B, t, err)) return false;
    out.variant = t.variant;
    const float *const stft = t.stft;
    const float *const *dww = t.dww, *const *dwb = t.dwb, *const *pww = t.pww, *const *pwb = t.pwb, *const *pjw = t.pjw, *const *pjb = t.pjb;
    const float *const *sw = t.sw, *const *sbias = t.sbias, *const *lwi = t.lwi, *const *lwh = t.lwh, *const *lbi = t.lbi, *const *lbh = t.lbh;
    const float *const head_w = t.head_w, *const head_b = t.head_b;
    StreamBuilder sb;
    uint32_t sec[S_COUNT] = {};
    auto dw_table = [&](const float *w5, const float *b, int C, int, int nquads) { return v4_dw_table(sb, w5, b, C, nquads); };
    v4_first_layer(sb, t, sec, v4_bin32);
    // S_S0: 16 -> 16 (rows 0..15)
    sec[S_S0] = sb.blocks();
    sb.vector_blocks([&](int c) { return c < 16 ? sbias[0][c] : 0.f; });
    for (int j = 0; j < 2; ++j) sb.weight_block([&](int np, int c) { return np < 16 ? sw[0][(size_t)np * 16 + c] : 0.f; }, j);
    // S_L1: dw table (4 quads), bias, pw (2 it), proj (2 it): 16 -> 32
    sec[S_L1] = dw_table(dww[1], dwb[1], 16, 0, 4);
    sb.vector_blocks([&](int c) { return pwb[1][c] + pjb[1][c]; });
    for (int j = 0; j < 2; ++j) sb.weight_block([&](int np, int c) { return pww[1][(size_t)np * 16 + c]; }, j);
    for (int j = 0; j < 2; ++j) sb.weight_block([&](int np, int c) { return pjw[1][(size_t)np * 16 + c]; }, j);
    // S_S1: 32 -> 32
    sec[S_S1] = sb.blocks();
    sb.vector_blocks([&](int c) { return sbias[1][c]; });
    for (int j = 0; j < 4; ++j) sb.weight_block([&](int np, int c) { return sw[1][(size_t)np * 32 + c]; }, j);
    // S_L2: dw table (8 quads), bias (pw only: identity residual), pw (4 it)
    sec[S_L2] = dw_table(dww[2], dwb[2], 32, 0, 8);
    sb.vector_blocks([&](int c) { return pwb[2][c]; });
    for (int j = 0; j < 4; ++j) sb.weight_block([&](int np, int c) { return pww[2][(size_t)np * 32 + c]; }, j);
    // S_S2
    sec[S_S2] = sb.blocks();
    sb.vector_blocks([&](int c) { return sbias[2][c]; });
    for (int j = 0; j < 4; ++j) sb.weight_block([&](int np, int c) { return sw[2][(size_t)np * 32 + c]; }, j);
    // S_L3: dw table (8 quads), then per output tile nt: bias, pw (4 it), proj (4 it): 32 -> 64
    sec[S_L3] = dw_table(dww[3], dwb[3], 32, 0, 8);
    for (int nt = 0; nt < 2; ++nt) {
        sb.vector_blocks([&](int c) { return pwb[3][32 * nt + c] + pjb[3][32 * nt + c]; });
        for (int j = 0; j < 4; ++j) sb.weight_block([&](int np, int c) { return pww[3][(size_t)(32 * nt + np) * 32 + c]; }, j);
        for (int j = 0; j < 4; ++j) sb.weight_block([&](int np, int c) { return pjw[3][(size_t)(32 * nt + np) * 32 + c]; }, j);
    }
    // S_S3: per output tile: bias, 8 it: 64 -> 64
    sec[S_S3] = sb.blocks();
    for (int nt = 0; nt < 2; ++nt) {
        sb.vector_blocks([&](int c) { return sbias[3][32 * nt + c]; });
        for (int j = 0; j < 8; ++j) sb.weight_block([&](int np, int c) { return sw[3][(size_t)(32 * nt + np) * 64 + c]; }, j);
    }
    // S_LSTM{0,1}: per wave w = hidden units 16w .. 16w+15, all four gates, full K (no partial sums to exchange):
    //   tile A rows 0..15 = gate i, rows 16..31 = gate f;  tile B rows = gates g | o   (PyTorch row order i,f,g,o)
    //   stream: bias A (4 blocks), bias B (4), then 16 k-iterations x {A, B}: iterations 0..7 contract the layer input
    //   (W_ih), 8..15 contract h_{t-1} (W_hh).  The MFMA D layout then puts i, f, g, o of a unit into the same lane.
    uint32_t lstm_sec[2][NWAVES];
    for (int l = 0; l < 2; ++l)
        for (int w = 0; w < NWAVES; ++w) {
            lstm_sec[l][w] = sb.blocks();
            auto grow = [&](int tile, int np) { return (2 * tile + (np >> 4)) * 64 + 16 * w + (np & 15); };   // gate row in [256]
            for (int tile = 0; tile < 2; ++tile)
                sb.vector_blocks([&](int c) { const int r = grow(tile, c); return lbi[l][r] + lbh[l][r]; });
            for (int j = 0; j < 16; ++j)
                for (int tile = 0; tile < 2; ++tile) {
                    const float *wsrc = j < 8 ? lwi[l] : lwh[l];
                    sb.weight_block([&](int np, int c) { return wsrc[(size_t)grow(tile, np) * 64 + c]; }, j & 7);
                }
        }
    // S_HEADB: block 0 float 0 = head bias; then head weights per unit half (4 vector blocks each)
    sec[S_HEADB] = sb.blocks();
    sb.new_block()[0] = head_b[0];
    for (int u = 0; u < 2; ++u) sb.vector_blocks([&](int c) { return head_w[32 * u + c]; });
    // the window of the stored basis: its k = 0 cosine row (the kernel multiplies by it while it folds)
    sec[S_NYQ] = sb.blocks();
    std::memcpy(sb.new_block(), stft, 256 * sizeof(float));
    for (int w = 0; w < NWAVES; ++w) {
        for (int k = 0; k < S_COUNT; ++k) out.sect[w][k] = sec[k];
        out.sect[w][S_LSTM0] = lstm_sec[0][w];
        out.sect[w][S_LSTM1] = lstm_sec[1][w];
        out.sect[w][S_STFT] = sb.blocks();
        pack_dft4_wave(sb, w);
    }
    out.data = std::move(sb.data);
    return true;
}

// ---- Silero V4 for the 16-stream tile kernel (csrc/silero_v4_t16.hip) ----------------------------------------------------------
// Every contraction on v_mfma_f32_16x16x4_f32 (blocks as in pack_silero_v5_t16: weight_block16 = 16 rows x 16 channels, a
// per-channel vector of a 16-row tile = one vector_block16).  A layer with C outputs has C / 16 row tiles rt; sections that every
// wave reads are packed once:
//   S_DW0, S_L0 : as pack_silero_v4 (the first layer runs on these tiles there, too)
//   S_S0  : bias, W                                   (16 -> 16, K = 16)
//   S_L1  : dw table, then per rt (2): bias, pw, proj (16 -> 32)
//   S_S1  : per rt (2): bias, 2 k-iterations          (32 -> 32)
//   S_L2  : dw table, per rt (2): bias, 2 k-it pw     (32 -> 32, identity residual)
//   S_S2  : per rt (2): bias, 2 k-it
//   S_L3  : dw table, per rt (4): bias, pw 2 k-it, proj 2 k-it   (32 -> 64)
//   S_S3  : per rt (4): bias, 4 k-it                  (64 -> 64)
//   S_LSTM{0,1} : per wave w (hidden units 16 w .. 16 w + 15): bias of gates i, f, g, o (4 blocks), then 8 k-iterations (0..3
//                 contract the layer input, 4..7 h_{t-1}) x 4 gates - the D layout puts the four gates of a unit into one lane
//   S_HEADB : head bias, then the head weights of wave w's units (block 1 + w);  S_NYQ : window;  S_STFT : as V5's t16 stream
bool pack_silero_v4_t16(const void *blob, size_t len, PackedWeights &out, std::string &err) {
    using namespace v4;
    Blob B;
    V4Tensors t;
    if (!load_v4(blob, len, B, t, err)) return false;
    out.variant = t.variant;
    StreamBuilder sb;
    uint32_t sec[S_COUNT] = {};
    if (!check_even_bin_fold(err)) return false;
    v4_first_layer(sb, t, sec, [](int c) { return c < 128 ? v4::bin_of_channel_t16(c) : c; });
    auto bias16 = [&](const float *a, const float *b2, int rt) {
        sb.vector_block16([&](int c) { return a[16 * rt + c] + (b2 ? b2[16 * rt + c] : 0.f); });
    };
    auto mat16 = [&](const float *W, int K, int rt, int j) {
        sb.weight_block16([&](int r, int c) { return W[(size_t)(16 * rt + r) * K + c]; }, j);
    };
    sec[S_S0] = sb.blocks();
    bias16(t.sbias[0], nullptr, 0);
    mat16(t.sw[0], 16, 0, 0);
    sec[S_L1] = v4_dw_table(sb, t.dww[1], t.dwb[1], 16, 4);
    for (int rt = 0; rt < 2; ++rt) {
        bias16(t.pwb[1], t.pjb[1], rt);
        mat16(t.pww[1], 16, rt, 0);
        mat16(t.pjw[1], 16, rt, 0);
    }
    sec[S_S1] = sb.blocks();
    for (int rt = 0; rt < 2; ++rt) {
        bias16(t.sbias[1], nullptr, rt);
        for (int j = 0; j < 2; ++j) mat16(t.sw[1], 32, rt, j);
    }
    sec[S_L2] = v4_dw_table(sb, t.dww[2], t.dwb[2], 32, 8);
    for (int rt = 0; rt < 2; ++rt) {
        bias16(t.pwb[2], nullptr, rt);
        for (int j = 0; j < 2; ++j) mat16(t.pww[2], 32, rt, j);
    }
    sec[S_S2] = sb.blocks();
    for (int rt = 0; rt < 2; ++rt) {
        bias16(t.sbias[2], nullptr, rt);
        for (int j = 0; j < 2; ++j) mat16(t.sw[2], 32, rt, j);
    }
    sec[S_L3] = v4_dw_table(sb, t.dww[3], t.dwb[3], 32, 8);
    for (int rt = 0; rt < 4; ++rt) {
        bias16(t.pwb[3], t.pjb[3], rt);
        for (int j = 0; j < 2; ++j) mat16(t.pww[3], 32, rt, j);
        for (int j = 0; j < 2; ++j) mat16(t.pjw[3], 32, rt, j);
    }
    sec[S_S3] = sb.blocks();
    for (int rt = 0; rt < 4; ++rt) {
        bias16(t.sbias[3], nullptr, rt);
        for (int j = 0; j < 4; ++j) mat16(t.sw[3], 64, rt, j);
    }
    uint32_t lstm_sec[2][NWAVES];
    for (int l = 0; l < 2; ++l)
        for (int w = 0; w < NWAVES; ++w) {
            lstm_sec[l][w] = sb.blocks();
            for (int q = 0; q < 4; ++q)
                sb.vector_block16([&](int c) { const int r = q * 64 + 16 * w + c; return t.lbi[l][r] + t.lbh[l][r]; });
            for (int j = 0; j < 8; ++j)
                for (int q = 0; q < 4; ++q) {
                    const float *wsrc = j < 4 ? t.lwi[l] : t.lwh[l];
                    sb.weight_block16([&](int r, int c) { return wsrc[(size_t)(q * 64 + 16 * w + r) * 64 + c]; }, j & 3);
                }
        }
    sec[S_HEADB] = sb.blocks();
    sb.new_block()[0] = t.head_b[0];
    for (int w = 0; w < NWAVES; ++w) sb.vector_block16([&](int c) { return t.head_w[16 * w + c]; });
    sec[S_NYQ] = sb.blocks();
    std::memcpy(sb.new_block(), t.stft, 256 * sizeof(float));
    const double two_pi = 6.283185307179586476925286766559;
    for (int w = 0; w < NWAVES; ++w) {
        for (int k = 0; k < S_COUNT; ++k) out.sect[w][k] = sec[k];
        out.sect[w][S_LSTM0] = lstm_sec[0][w];
        out.sect[w][S_LSTM1] = lstm_sec[1][w];
        out.sect[w][S_STFT] = sb.blocks();
        pack_dft_fold3_wave(sb, w);
    }
    out.data = std::move(sb.data);
    return true;
}

}  // namespace vadk

// Host-side weight packing (see pack_weights.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "vad_layout.h"

namespace vadk {

struct PackedWeights {
    std::vector<float> data;        // concatenated per-wave streams, multiple of BLK_FLOATS
    uint32_t sect[NWAVES][16] = {};  // block offset of every section
    int32_t variant = 0;             // V4: 1 = the graph's 8 kHz sub-model (two time steps reach the LSTMs)
};

// blob: SVW container (cutter_vad_amd/weights_io.py).  On failure returns false and sets err
// to a message starting with "Failed to load model" (matches the reference's
// ModelInitializationError text, core/silero_model.py:330-334).
bool pack_silero_v5(const void *blob, size_t len, PackedWeights &out, std::string &err);
bool pack_silero_v4(const void *blob, size_t len, PackedWeights &out, std::string &err);
// Silero V4 (either sub-model) repacked for the 16-stream tile kernel (csrc/silero_v4_t16.hip)
bool pack_silero_v4_t16(const void *blob, size_t len, PackedWeights &out, std::string &err);
// Silero V5 16 kHz repacked for the 16-stream tile kernel (16 x 16 x 4 MFMA tiles; csrc/silero_v5_t16.hip)
bool pack_silero_v5_t16(const void *blob, size_t len, PackedWeights &out, std::string &err);

// scipy.signal.resample(x, 512) for len(x) == n_in as a dense operator R[512][n_in] (row-major),
// built in double precision from the closed form of the Fourier method (utils/audio.py:46-49).
void build_resample_operator(int n_in, std::vector<float> &R);
// radix-2 + mirror-folded packing of that operator (pack_weights.cpp); returns the blocks per row tile, 0 on failure
uint32_t pack_resample_operator(int n_in, std::vector<float> &out, uint32_t *row128_block, std::string &err);
// the same operator packed for the fused resample -> V5 kernel on 16-stream tiles; returns the blocks per wave, 0 on failure
uint32_t pack_resample_operator_t16(int n_in, std::vector<float> &out, uint32_t *row128_block, std::string &err);

}  // namespace vadk

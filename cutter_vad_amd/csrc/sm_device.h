// Device-side hysteresis state machine shared by every model kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "vad_layout.h"

namespace vadk {

// ---- hysteresis state machine, one stream, one frame (core/silero_model.py:790-949) ----
__device__ __forceinline__ int sm_step(SmSlot &s, float p, int *seg_out) {
    int ev = 0;
    if (!s.active) {
        const bool above = (double)p >= s.start_prob;                                  // :832
        s.start_hist = (s.start_hist << 1) | (above ? 1u : 0u);                // :833 deque(maxlen=20)
        s.start_len = min(s.start_len + 1, 20);
        if (above) {
            s.n_start += 1;                                                    // :836
            s.buffered += 1;                                                   // :839
            if (s.n_start >= s.start_count && s.start_len >= s.start_count) {  // :842-843
                const int k = s.start_count;
                const uint32_t mask = k >= 32 ? 0xffffffffu : ((1u << k) - 1u);
                const double ratio = (double)__popc(s.start_hist & mask) / (double)k; // :846-848
                if (ratio >= s.start_ratio) {                                  // :851
                    s.active = 1; s.n_start = 0; s.n_end = 0;                  // :862-864
                    s.seg_frames = s.buffered > 0 ? s.buffered : -1;           // :867-868
                    s.buffered = 0;                                            // :869
                    ev |= 1;
                }
            }
        } else {
            s.n_start = 0; s.buffered = 0;                                     // :873-874
        }
    } else {
        s.seg_frames = (s.seg_frames < 0 ? 0 : s.seg_frames) + 1;              // :891, :925-930
        ev |= 4;                                                               // :894
        const bool below = (double)p < s.end_prob;                                     // :898
        // :899 deque(maxlen=100): 128-bit shift register, bits above 100 are never read
        s.end_hist[3] = (s.end_hist[3] << 1) | (s.end_hist[2] >> 31);
        s.end_hist[2] = (s.end_hist[2] << 1) | (s.end_hist[1] >> 31);
        s.end_hist[1] = (s.end_hist[1] << 1) | (s.end_hist[0] >> 31);
        s.end_hist[0] = (s.end_hist[0] << 1) | (below ? 1u : 0u);
        s.end_len = min(s.end_len + 1, 100);
        if (below) {
            s.n_end += 1;                                                      // :903
            if (s.n_end >= s.end_count && s.end_len >= s.end_count) {          // :906-907
                const int k = s.end_count;
                int cnt = 0;
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    const int lo = wd * 32;
                    if (k > lo) {
                        const int nb = min(k - lo, 32);
                        const uint32_t mask = nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u);
                        cnt += __popc(s.end_hist[wd] & mask);
                    }
                }
                const double ratio = (double)cnt / (double)k;                   // :910-912
                if (ratio >= s.end_ratio) {                                    // :915
                    *seg_out = s.seg_frames < 0 ? 0 : s.seg_frames;            // :941-942
                    s.active = 0; s.n_end = 0; s.seg_frames = -1;              // :945-947
                    ev |= 2;
                }
            }
        } else {
            s.n_end = 0;                                                       // :921
        }
    }
    return ev;
}


}  // namespace vadk

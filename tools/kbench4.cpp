// Developer micro-benchmark for the V4 kernel pair (not part of the product, not a test).
//   tools/kbench4.sh [-DVADK_STAMPS] -- <B> <steps> [8k]
// Times back-to-back steps (2 launches each) with hipEvents; with -DVADK_STAMPS both kernels record s_memtime at their
// phase boundaries (STFT part: slots 0..15, tail: slots 16..31) and this program prints the per-phase cycle budget.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../cutter_vad_amd/csrc/pack_weights.h"
#include "../cutter_vad_amd/csrc/vad_layout.h"

extern "C" hipError_t vadk_launch_silero_v4(const vadk::StepParams *p, hipStream_t stream);
#ifdef KB_TILE16      // the 16-stream tile kernel (silero_v4_t16.hip); KB_ONE_PER_CU: one workgroup per CU
extern "C" hipError_t vadk_launch_silero_v4_t16(const vadk::StepParams *p, int one_per_cu, hipStream_t stream);
#ifndef KB_ONE_PER_CU
#define KB_ONE_PER_CU 0
#endif
#define vadk_launch_silero_v4(p, s) vadk_launch_silero_v4_t16((p), KB_ONE_PER_CU, (s))
#define KB_MT 16
#define KB_PACK vadk::pack_silero_v4_t16
#else
#define KB_MT vadk::MT
#define KB_PACK vadk::pack_silero_v4
#endif

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

int main(int argc, char **argv) {
    const char *blob_path = argc > 1 ? argv[1] : "cutter_vad_amd/weights/silero_v4_16k.svw";
    const int B = argc > 2 ? atoi(argv[2]) : 8192;
    const int K = argc > 3 ? atoi(argv[3]) : 100;
    FILE *f = fopen(blob_path, "rb");
    if (!f) { perror(blob_path); return 1; }
    std::vector<unsigned char> blob;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
    fclose(f);
    vadk::PackedWeights pw;
    std::string err;
    if (!KB_PACK(blob.data(), blob.size(), pw, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const int tiles = (B + KB_MT - 1) / KB_MT;
    vadk::StepParams p{};
    float *d_w, *d_state, *d_frames, *d_probs;
    vadk::SmSlot *d_sm;
    CK(hipMalloc(&d_w, pw.data.size() * 4));
    CK(hipMemcpy(d_w, pw.data.data(), pw.data.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_state, (size_t)B * 256 * 4));
    CK(hipMemset(d_state, 0, (size_t)B * 256 * 4));
    std::vector<vadk::SmSlot> sm(B);
    for (auto &s : sm) { memset(&s, 0, sizeof s); s.start_prob = s.end_prob = 0.7; s.start_ratio = 0.8; s.end_ratio = 0.95; s.start_count = 10; s.end_count = 50; s.seg_frames = -1; }
    CK(hipMalloc(&d_sm, sizeof(vadk::SmSlot) * B));
    CK(hipMemcpy(d_sm, sm.data(), sizeof(vadk::SmSlot) * B, hipMemcpyHostToDevice));
    const int RING = 16;
    std::vector<float> fr((size_t)RING * B * 512);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 0.1f);
    for (auto &v : fr) v = nd(rng);
    CK(hipMalloc(&d_frames, fr.size() * 4));
    CK(hipMemcpy(d_frames, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_probs, (size_t)B * 4));
    p.wstream = d_w;
    p.wstream_bytes = (uint32_t)(pw.data.size() * 4);
    memcpy(p.sect, pw.sect, sizeof pw.sect);
    p.variant = pw.variant;
    p.state = d_state; p.sm = d_sm; p.slots = nullptr; p.probs = d_probs; p.events = nullptr; p.seg_frames = nullptr;
    p.n = B; p.T = 1; p.fmt = 0; p.thresh = 0.01f;
#ifdef VADK_STAMPS
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)tiles * 4 * 32 * 8));
    CK(hipMemset(d_st, 0, (size_t)tiles * 4 * 32 * 8));
    p.stamps = d_st;
#endif
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 5; ++i) { p.frames = d_frames + (size_t)(i % RING) * B * 512; CK(vadk_launch_silero_v4(&p, s)); }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> rounds;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < K; ++i) { p.frames = d_frames + (size_t)(i % RING) * B * 512; CK(vadk_launch_silero_v4(&p, s)); }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        rounds.push_back(ms * 1e3f / K);
    }
    std::sort(rounds.begin(), rounds.end());
    const double us = rounds[rounds.size() / 2];
    printf("V4%s B=%d: median %.1f us/step (min %.1f)  %.2f M frames/s\n", pw.variant ? "-8k" : "", B, us, rounds[0], B / us);
#ifdef VADK_STAMPS
    std::vector<unsigned long long> st((size_t)tiles * 4 * 32);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    auto S = [&](int b, int w, int k) { return st[((size_t)b * 4 + w) * 32 + k]; };
    // cycles between consecutive stamps that exist (a stamp index that the kernel does not write is skipped)
    for (int w = 0; w < 4; ++w) {
        printf("wave %d:", w);
        int prev = -1;
        for (int k = 0; k < 32; ++k) {
            int have = 0;
            for (int b = 0; b < tiles; ++b) have += S(b, w, k) != 0;
            if (!have) continue;
            if (prev >= 0) {
                double acc = 0; int cnt = 0;
                for (int b = 0; b < tiles; ++b) if (S(b, w, k) && S(b, w, prev)) { acc += (double)(long long)(S(b, w, k) - S(b, w, prev)); ++cnt; }
                if (cnt) printf(" %d>%d=%.0f", prev, k, acc / cnt);
            }
            prev = k;
        }
        printf("\n");
    }
#endif
    return 0;
}

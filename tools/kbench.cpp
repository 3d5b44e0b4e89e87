// Developer micro-benchmark for the fused V5 kernel (not part of the product, not a test).
//
//   tools/kbench.sh [-DVADK_STAMPS] -- <B> <steps>
//
// Builds silero_v5.hip + pack_weights.cpp directly (so kernel variants can be compiled with
// extra -D flags on the GPU box) and times back-to-back launches with hipEvents.  With
// -DVADK_STAMPS the kernel records s_memtime at every phase boundary per wave and this program
// prints the per-phase cycle budget (shares only: stamped builds are slower, see
// cdna_hip_programming.md "In-kernel stamps").
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

extern "C" hipError_t vadk_launch_silero_v5(const vadk::StepParams *p, hipStream_t stream);
#ifdef KB_TILE16      // tools/kbench16.sh: the 16-stream tile kernel (link silero_v5_t16.hip instead of silero_v5.hip)
extern "C" hipError_t vadk_launch_silero_v5_t16(const vadk::StepParams *p, hipStream_t stream);
#define vadk_launch_silero_v5 vadk_launch_silero_v5_t16
#define PACK vadk::pack_silero_v5_t16
#else
#define PACK vadk::pack_silero_v5
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
    const char *blob_path = argc > 1 ? argv[1] : "cutter_vad_amd/weights/silero_v5_16k.svw";
    const int B = argc > 2 ? atoi(argv[2]) : 8192;
    const int K = argc > 3 ? atoi(argv[3]) : 100;
    const int T = argc > 4 ? atoi(argv[4]) : 1;
    FILE *f = fopen(blob_path, "rb");
    if (!f) { perror(blob_path); return 1; }
    std::vector<unsigned char> blob;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
    fclose(f);
    vadk::PackedWeights pw;
    std::string err;
    if (!PACK(blob.data(), blob.size(), pw, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }

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
    const int RING = getenv("KB_RING") ? atoi(getenv("KB_RING")) : 8;
    const float sigma = getenv("KB_SIGMA") ? (float)atof(getenv("KB_SIGMA")) : 0.1f;
    std::vector<float> fr((size_t)RING * B * T * 512);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, sigma);
    for (auto &v : fr) v = nd(rng);
    CK(hipMalloc(&d_frames, fr.size() * 4));
    CK(hipMemcpy(d_frames, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_probs, (size_t)B * T * 4));
    p.wstream = d_w;
    p.wstream_bytes = (uint32_t)(pw.data.size() * 4);
    memcpy(p.sect, pw.sect, sizeof pw.sect);
    p.state = d_state; p.sm = d_sm; p.slots = nullptr; p.probs = d_probs; p.events = nullptr; p.seg_frames = nullptr;
    p.n = B; p.T = T; p.fmt = 0; p.thresh = 0.01f;
    if (getenv("KB_SLOTS")) {          // identity slot map, like vad_step_device
        std::vector<int> sl(B);
        for (int i = 0; i < B; ++i) sl[i] = i;
        int *d_sl; CK(hipMalloc(&d_sl, B * 4)); CK(hipMemcpy(d_sl, sl.data(), B * 4, hipMemcpyHostToDevice));
        p.slots = d_sl;
    }
    if (getenv("KB_EVENTS")) { unsigned char *d_ev; CK(hipMalloc(&d_ev, (size_t)B * T)); p.events = d_ev; }
#ifdef VADK_STAMPS
#ifdef KB_TILE16
    const int tiles = (B + 15) / 16;
#else
    const int tiles = (B + vadk::MT - 1) / vadk::MT;
#endif
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)tiles * 4 * 32 * 8));
    CK(hipMemset(d_st, 0, (size_t)tiles * 4 * 32 * 8));
    p.stamps = d_st;
#endif
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 5; ++i) { p.frames = d_frames + (size_t)(i % RING) * B * T * 512; CK(vadk_launch_silero_v5(&p, s)); }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> rounds;
    // KB_SPLIT=n: the B streams as n independent groups, each stepped K times on a HIP stream of its own - the groups' launches are
    // not ordered against each other, so the start of one group's launch runs under the other groups' tiles
    const int NS = getenv("KB_SPLIT") ? atoi(getenv("KB_SPLIT")) : 1;
    std::vector<hipStream_t> ss(NS, s);
    std::vector<hipEvent_t> ej(NS);
    for (int g = 1; g < NS; ++g) CK(hipStreamCreate(&ss[g]));
    for (int g = 0; g < NS; ++g) CK(hipEventCreate(&ej[g]));
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        if (NS == 1) {
            for (int i = 0; i < K; ++i) { p.frames = d_frames + (size_t)(i % RING) * B * T * 512; CK(vadk_launch_silero_v5(&p, s)); }
        } else {
            const int per = B / NS;
            static std::vector<float *> wcopy;           // KB_SPLIT_OWN_WEIGHTS: a copy of the weight stream per group (two engines)
            if (getenv("KB_SPLIT_OWN_WEIGHTS") && wcopy.empty()) {
                wcopy.assign(NS, d_w);
                for (int g = 1; g < NS; ++g) {
                    CK(hipMalloc(&wcopy[g], pw.data.size() * 4));
                    CK(hipMemcpy(wcopy[g], d_w, pw.data.size() * 4, hipMemcpyDeviceToDevice));
                }
            }
            for (int g = 1; g < NS; ++g) CK(hipStreamWaitEvent(ss[g], e0, 0));
            for (int i = 0; i < K; ++i)
                for (int g = 0; g < NS; ++g) {
                    vadk::StepParams q = p;
                    q.n = per;
                    q.state = p.state + (size_t)g * per * 256;
                    q.sm = p.sm + (size_t)g * per;
                    q.probs = p.probs + (size_t)g * per * T;
                    q.frames = d_frames + ((size_t)(i % RING) * B + (size_t)g * per) * T * 512;
                    if (!wcopy.empty()) q.wstream = wcopy[g];
                    CK(vadk_launch_silero_v5(&q, ss[g]));
                }
            for (int g = 1; g < NS; ++g) { CK(hipEventRecord(ej[g], ss[g])); CK(hipStreamWaitEvent(s, ej[g], 0)); }
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        rounds.push_back(ms * 1e3f / K);
    }
    std::sort(rounds.begin(), rounds.end());
    const double us = rounds[rounds.size() / 2];
    const double fps = (double)B * T / (us * 1e-6);
    printf("B=%d T=%d: median %.1f us/launch (min %.1f)  %.2f M frames/s  %.1f TFLOP/s = %.3f of fp32 peak\n", B, T, us,
           rounds[0], fps / 1e6, fps * 988160 / 1e12, fps * 988160 / 157.3e12);
#ifdef VADK_STAMPS
    std::vector<unsigned long long> st((size_t)tiles * 4 * 32);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    static const char *names[16] = {"", "load", "bar1", "stft", "bar2", "enc0", "bar3", "enc1", "bar4", "enc2", "bar5", "enc3", "bar6", "lstm", "bar7", "cell+bar8"};
    auto S = [&](int b, int w, int k) { return st[((size_t)b * 4 + w) * 32 + k]; };
    for (int w = 0; w < 4; ++w) {
        printf("wave %d:", w);
        double tot = 0;
        for (int k = 1; k < 16; ++k) {
            double acc = 0;
            for (int b = 0; b < tiles; ++b) acc += (double)(S(b, w, k) - S(b, w, k - 1));
            acc /= tiles;
            tot += acc;
            printf(" %s=%.0f", names[k], acc);
        }
        double l1 = 0, l2 = 0, l3 = 0;
        for (int b = 0; b < tiles; ++b) { l1 += (double)(S(b, w, 16) - S(b, w, 2)); l2 += (double)(S(b, w, 17) - S(b, w, 4)); l3 += (double)(S(b, w, 18) - S(b, w, 12)); }
        printf("  total=%.0f | mainloops: stft=%.0f enc0=%.0f lstm=%.0f\n", tot, l1 / tiles, l2 / tiles, l3 / tiles);
        printf("        ingest groups (cycles since frame start):");
        for (int k = 20; k < 28; ++k) {
            double acc = 0;
            for (int b = 0; b < tiles; ++b) acc += (double)(S(b, w, k) - S(b, w, 0));
            printf(" g%d=%.0f", k - 20, acc / tiles);
        }
        double pro = 0;
        for (int b = 0; b < tiles; ++b) pro += (double)(S(b, w, 0) - S(b, w, 19));
        printf("   prologue (kernel entry -> frame loop) = %.0f\n", pro / tiles);
        // (s_memtime is a per-CU counter: stamps of different tiles cannot be set against each other)
        double a29 = 0, a30 = 0, a31 = 0;
        for (int b = 0; b < tiles; ++b) {
            a29 += (double)(S(b, w, 29) - S(b, w, 19)); a30 += (double)(S(b, w, 30) - S(b, w, 19)); a31 += (double)(S(b, w, 31) - S(b, w, 19));
        }
        if (w == 0 && S(0, 0, 28)) {
            double tl = 0;
            for (int b = 0; b < tiles; ++b) tl += (double)(S(b, 0, 28) - S(b, 0, 15));
            printf("        head + state machine (32 lanes of wave 0, behind barrier (8)) = %.0f\n", tl / tiles);
        }
        if (S(0, w, 29)) printf("        since kernel entry: all first requests issued %.0f, h in LDS %.0f, past barrier (0) %.0f\n", a29 / tiles, a30 / tiles, a31 / tiles);
    }
#endif
    return 0;
}

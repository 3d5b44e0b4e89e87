// ASan/UBSan harness for the host-side packers: see tools/sanitize_packer.sh
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#include "pack_weights.h"
static std::vector<char> rd(const char *p) { std::ifstream f(p, std::ios::binary); return std::vector<char>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char **argv) {
    std::string err;
    for (int i = 1; i < argc; ++i) {
        auto b = rd(argv[i]);
        vadk::PackedWeights pw;
        bool ok = std::string(argv[i]).find("v5") != std::string::npos ? vadk::pack_silero_v5(b.data(), b.size(), pw, err) : vadk::pack_silero_v4(b.data(), b.size(), pw, err);
        printf("%s: %d %zu %s\n", argv[i], (int)ok, pw.data.size(), err.c_str());
        {   // the 16-stream tile packings of the same blob (V5's exists for the 16 kHz sub-model only)
            vadk::PackedWeights p16; std::string e16;
            const bool v5 = std::string(argv[i]).find("v5") != std::string::npos;
            const bool ok16 = v5 ? vadk::pack_silero_v5_t16(b.data(), b.size(), p16, e16) : vadk::pack_silero_v4_t16(b.data(), b.size(), p16, e16);
            printf("  t16: %d %zu %s\n", (int)ok16, p16.data.size(), e16.c_str());
            for (size_t cut : {(size_t)0, (size_t)64, b.size() / 2, b.size() - 1}) {
                vadk::PackedWeights q; std::string e2;
                if (v5 ? vadk::pack_silero_v5_t16(b.data(), cut, q, e2) : vadk::pack_silero_v4_t16(b.data(), cut, q, e2)) printf("  t16 cut %zu unexpectedly ok\n", cut);
            }
        }
        // truncated / corrupted blobs must fail cleanly
        for (size_t cut : {(size_t)0, (size_t)7, (size_t)64, b.size() / 2, b.size() - 1}) {
            vadk::PackedWeights q; std::string e2;
            bool k2 = std::string(argv[i]).find("v5") != std::string::npos ? vadk::pack_silero_v5(b.data(), cut, q, e2) : vadk::pack_silero_v4(b.data(), cut, q, e2);
            if (k2) printf("  cut %zu unexpectedly ok\n", cut);
        }
        // a tensor-table entry whose offset is near 2^64 (offset + 4 * nelem wraps to a small number) must be refused, not read
        if (b.size() > 16 + 88) {
            std::vector<char> evil(b);
            const unsigned long long off = ~0ull - 1023;       // entry 0 starts at byte 16; its offset field follows name[48], ndim, dims[4], reserved = byte 72
            std::memcpy(evil.data() + 16 + 72, &off, 8);
            vadk::PackedWeights q; std::string e2;
            bool k2 = std::string(argv[i]).find("v5") != std::string::npos ? vadk::pack_silero_v5(evil.data(), evil.size(), q, e2) : vadk::pack_silero_v4(evil.data(), evil.size(), q, e2);
            printf("  wrapped offset: %s (%s)\n", k2 ? "UNEXPECTEDLY OK" : "refused", e2.c_str());
        }
    }
    for (int n : {256, 768, 1536}) {
        std::vector<float> out; uint32_t r128 = 0;
        uint32_t tb = vadk::pack_resample_operator(n, out, &r128, err);
        printf("resample %d: %u %u %zu\n", n, tb, r128, out.size());
        std::vector<float> o16; uint32_t r16 = 0;
        uint32_t wb = vadk::pack_resample_operator_t16(n, o16, &r16, err);
        printf("resample t16 %d: %u %u %zu\n", n, wb, r16, o16.size());
    }
    return 0;
}

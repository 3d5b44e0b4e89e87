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
    }
    return 0;
}

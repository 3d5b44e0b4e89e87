// ASan/UBSan harness for the host-side packers: see tools/sanitize_packer.sh
#include <cstdio>
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
    }
    for (int n : {256, 768, 1536}) {
        std::vector<float> out; uint32_t r128 = 0;
        uint32_t tb = vadk::pack_resample_operator(n, out, &r128, err);
        printf("resample %d: %u %u %zu\n", n, tb, r128, out.size());
    }
    return 0;
}

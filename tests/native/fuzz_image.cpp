// Sanitizer fuzz driver for the image decoders (tests/test_native_sanitizers.py): mutates a seed PNG / JPEG (byte flips, 16- and
// 32-bit overwrites, truncation) and decodes every mutant. Built with g++ -fsanitize=address,undefined; usage: fuzz SEED ITERS RNG.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/sunray_hip.h"
namespace srh { int set_error(int code, const std::string&) { return code; } }
int main(int argc, char** argv) {
    std::vector<unsigned char> seed;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); seed.resize(n); fread(seed.data(), 1, n, f); fclose(f);
    unsigned iters = atoi(argv[2]), rng = atoi(argv[3]);
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; };
    unsigned ok = 0;
    std::vector<unsigned char> out(1 << 22);
    for (unsigned it = 0; it < iters; it++) {
        std::vector<unsigned char> m = seed;
        unsigned k = it == 0 ? 0 : 1 + next() % 6;
        for (unsigned j = 0; j < k; j++) {
            unsigned pos = next() % m.size();
            switch (next() % 5) {
                case 0: m[pos] = (unsigned char)next(); break;
                case 1: m[pos] ^= 1u << (next() % 8); break;
                case 2: if (pos + 4 <= m.size()) { unsigned v = next() % 3 == 0 ? 0xFFFFFFFFu : next(); memcpy(&m[pos], &v, 4); } break;
                case 3: if (m.size() > 32) m.resize(m.size() - next() % 24); break;
                case 4: if (pos + 2 <= m.size()) { m[pos] = 0xFF; m[pos + 1] = (unsigned char)(0xC0 + next() % 0x40); } break;   // plant a marker
            }
        }
        uint32_t w = 0, h = 0, c = 0;
        if (sr_decode_image(m.data(), m.size(), &w, &h, &c, nullptr, 0) == 0 && (size_t)w * h * c <= out.size()) {
            if (sr_decode_image(m.data(), m.size(), &w, &h, &c, out.data(), out.size()) == 0) ok++;
        }
        if (sr_decode_image_rgba8(m.data(), m.size(), &w, &h, nullptr, 0) == 0 && (size_t)w * h * 4 <= out.size())      // + 16-bit PNG
            (void)sr_decode_image_rgba8(m.data(), m.size(), &w, &h, out.data(), out.size());
    }
    printf("%s: %u iterations, %u decoded\n", argv[1], iters, ok);
    return 0;
}
